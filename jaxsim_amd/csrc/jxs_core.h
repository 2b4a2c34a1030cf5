// Kernel core of the batched rigid-body step: one link per lane, one environment per lane
// group, wavefront shuffles along the kinematic tree.  Templated on the lane backend so the
// identical source runs as the gfx950 kernel (jxs_lanes_device.h) and as a host lockstep
// emulation in the CPU tests (tests/emul/jxs_lanes_host.h).  The code is branch-free over
// lane values: every branch below is wave-uniform (depends only on KParams / the mode).
//
// Formulation: the reference runs Featherstone's ABA in link
// coordinates with 6x6 Pluecker transforms between neighbours (`src/jaxsim/rbda/aba.py`).
// Here every spatial quantity is expressed in ONE frame C = (origin at the base-link
// position, world-aligned axes, at rest).  All neighbour transforms become the identity, so
// the serial sweeps shrink to "add the child's articulated inertia to the parent", and
// forward kinematics / velocities / RNEA accelerations become tree prefix sums done by
// pointer jumping in ceil(log2(depth+1)) shuffle rounds.  Spatial vectors are
// [linear; angular] like the reference (`src/jaxsim/math/adjoint.py:92-107`); in exact
// arithmetic the results equal the reference's.
//
// Reference rows implemented here:
//   B actuation   api/actuation_model.py:7-126      H joint transforms  api/kin_dyn_parameters.py:396-451
//   M kinematics  rbda/forward_kinematics.py:12-113 J,K,L soft contacts rbda/contacts/soft.py:195-444
//   I point->link api/contact.py:557-603            G ABA               rbda/aba.py:12-292
//   C integrator  api/integrators.py:14-88          R RNEA              rbda/rnea.py:12-238
//   A step glue   api/model.py:2601-2681            N cache refresh     api/data.py:405-523
//   RK4           api/integrators.py:91-167, api/ode.py:134-225 (MODE_STEP_RK4)
//   S5 rigid contacts rbda/contacts/rigid.py:176-539 (MODE_STEP_RIGID, member functions in jxs_rigid.inc)
#pragma once
#include <limits>
#include "jxs_params.h"

namespace jxs {

template <class L>
struct Core {
  using T = typename L::T;
  using V = typename L::V;
  using VI = typename L::VI;
  using VM = typename L::VM;
  static constexpr int G = L::G;
#ifdef JXS_NO_LANE_ROWS  // developer knob: A/B the table-independent loads of the joint / deformation rows (run(), stage A)
  static constexpr bool kLaneRows = false;
#else
  static constexpr bool kLaneRows = true;
#endif
#ifdef JXS_NO_ROW_BASE_SOLVE  // developer knob: A/B the base solve of the row-distributed sweeps (aba_rows)
  static constexpr bool kNoRowBaseSolve = true;
#else
  static constexpr bool kNoRowBaseSolve = false;
#endif

  const KParams<T>& P;
  const KArgs<T>& A;
  const L& ln;
  // rigid contact models: bit j set = point j is active in some environment of this wave (set by
  // rigid_delassus; block columns of the other points are skipped by the factorisation and the solves)
  using PointMask = unsigned long long;  // one bit per collidable point of the rigid contact models (<= 64)
  mutable PointMask rg_amask_ = ~0ull;
  mutable VM rg_mine_;  // this lane's own point has its bit set
  mutable int rg_hoff_ = 0;  // LDS word offset of the working matrix H: behind Q, or Q itself (relaxed model)
  // rigid contact models: active set the Delassus matrix in the LDS was last built for (rigid_contact_forces) and
  // whether it is there at all -- the impact reuses it as its preconditioner when the set has not changed
  mutable VM rg_act_last_;
  mutable bool rg_q_valid_ = false;
  mutable int duo_seen_ = 0;  // two-wave workgroups, main wave: last value read from the inertia wave's progress word

  JXS_HD Core(const KParams<T>& p, const KArgs<T>& a, const L& l) : P(p), A(a), ln(l) {}

  // ---- small dense helpers on lane values ------------------------------------------------
  static JXS_HD void cross(const V* a, const V* b, V* o) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
  }
  // acc[k] += x[k]@(lane + 1) where `ok`, k < N (6 or 9): a parent of the link-per-lane sweeps takes the share of its FIRST
  // child, which sits in the next lane (depth-first lane order).  Fused form: a DPP lane shift as the operand of a
  // multiply-add with the mask as a 0 / 1 factor (values are finite by construction) -- with the last lane of the group
  // switched off, because ITS next lane is the base link of the next environment and 0 x NaN of a diverged neighbour
  // would walk down the padding lanes into this one (tests: test_a_non_finite_environment_does_not_touch_its_neighbours).
  // That lane must be a padding lane for this (nobody needs its value); a model that fills its group takes the selects.
  template <int N>
  JXS_HD void add_from_next(const VI& lane, V* acc, const V* x, const VM& ok) const {
    static_assert(N == 6 || N == 9, "six or nine values");
    if (P.nL < G) {
      const V okf = vsel(ok, V(T(1)), V(T(0)));
      if (N == 9) ln.fmac9_from_next(acc, x, okf, lane < G - 1);
      else ln.fmac6_from_next(acc, x, okf, lane < G - 1);
    } else {
#pragma unroll
      for (int k = 0; k < N; ++k) acc[k] = acc[k] + vsel(ok, ln.from_next(x[k]), V(T(0)));
    }
  }
  static JXS_HD void mat3vec(const V* R, const V* x, V* o) {  // o = R x, R row-major
    o[0] = R[0] * x[0] + R[1] * x[1] + R[2] * x[2];
    o[1] = R[3] * x[0] + R[4] * x[1] + R[5] * x[2];
    o[2] = R[6] * x[0] + R[7] * x[1] + R[8] * x[2];
  }
  static JXS_HD void mat3mul(const V* a, const V* b, V* o) {
    if (L::mat3mul_packed(a, b, o)) return;  // device fp32: v_pk_fma_f32 (jxs_lanes_device.h)
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) o[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
  }
  // symmetric 6x6 stored as upper triangle, row-major: idx(i,j), i <= j
  static JXS_HD constexpr int sidx(int i, int j) {
    return (i <= j) ? (i * 6 - (i * (i - 1)) / 2 + (j - i)) : (j * 6 - (j * (j - 1)) / 2 + (i - j));
  }

  // Reference-point change of an articulated inertia and its bias force, both world-aligned: from a point
  // O_c to O_p with d = O_c - O_p.  M' = X^T M X, p' = X^T p with X = [[I, -S(d)], [0, I]] (motion about O_p
  // -> motion about O_c).  Blocks M = [[A, B], [B^T, D]]:
  //   A' = A,   B' = B - A S(d),   D' = D + S(d) B' + (S(d) B)^T,   p' = [f; n + d x f].
  static JXS_HD void xlate_inertia(V* M, V* p, const V* d) {
    V Y[3][3], Bn[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {  // Y_i = A_i x d  (row i of A S(d))
      const V a0 = M[sidx(i, 0)], a1 = M[sidx(i, 1)], a2 = M[sidx(i, 2)];
      Y[i][0] = a1 * d[2] - a2 * d[1];
      Y[i][1] = a2 * d[0] - a0 * d[2];
      Y[i][2] = a0 * d[1] - a1 * d[0];
    }
    // W = S(d) B and W' = S(d) B': entry [i][j] = (d x column j)_i
    auto sx = [&](const V (*Bm)[3], int i, int j) {
      const int i1 = (i + 1) % 3, i2 = (i + 2) % 3;
      return d[i1] * Bm[i2][j] - d[i2] * Bm[i1][j];
    };
    V Bo[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) Bo[i][j] = M[sidx(i, 3 + j)], Bn[i][j] = Bo[i][j] - Y[i][j];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = i; j < 3; ++j) M[sidx(3 + i, 3 + j)] = M[sidx(3 + i, 3 + j)] + sx(Bn, i, j) + sx(Bo, j, i);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) M[sidx(i, 3 + j)] = Bn[i][j];
    V t[3];
    cross(d, p, t);
#pragma unroll
    for (int k = 0; k < 3; ++k) p[3 + k] = p[3 + k] + t[k];
  }

  // ==========================================================================================
  // ROLE_SOLO: the whole step in one wave.  ROLE_MAIN: the main wave of a two-wave workgroup -- everything but the
  // articulated-inertia recursion of ABA pass 2, which the inertia wave (run_inertia) runs concurrently from the
  // same joint positions and hands over level by level (aba_rows_main).
  template <int MODE, int ROLE = ROLE_SOLO>
  JXS_HD void run() {
    static_assert(ROLE == ROLE_SOLO || (ROLE == ROLE_MAIN && MODE == MODE_STEP), "two-wave workgroups: single step only");
    constexpr int kArea = (ROLE == ROLE_MAIN) ? duo_main_off(G) : 0;  // LDS area of this wave (words)
    constexpr bool kRK4 = (MODE == MODE_STEP_RK4 || MODE == MODE_STEP_RK4_RIGID);
    // [round 6] system_dynamics / link_contact_forces (api/ode.py:16-225, api/contact.py:514-603): the step without the
    // actuation model and without the integrator; the derivative of the state and the link contact wrenches are the outputs
    constexpr bool kDyn = (MODE == MODE_DYN || MODE == MODE_DYN_RIGID);
    constexpr bool kRigid = (MODE == MODE_STEP_RIGID || MODE == MODE_STEP_RK4_RIGID || MODE == MODE_DYN_RIGID);
    constexpr bool kStep = (MODE == MODE_STEP || MODE == MODE_ROLLOUT || kRK4 || kRigid || kDyn);
    const VI lane = ln.lane();
    ln.stamp(A, 0);
    ln.stamp_hwid(A, 11);

    // ======================================================================================
    // Batch 1 of global loads: every address below is known at launch, so all of them are in
    // flight together (one memory round trip instead of one per phase; rocprof showed 44 % of the
    // wave cycles parked in s_waitcnt before this was hoisted).  Loads are unconditional -- rows
    // are clamped into range and the value masked afterwards -- so no exec-mask branches.
    // ======================================================================================
    // Stage A: everything whose address needs only the PRELOADED kernel arguments (table and state
    // pointers, row counts: SGPRs that arrive with the wave) is issued first and unconditionally, so that
    // these loads are in flight while the scalar loads of the rest of the argument block (KParams, the
    // other pointers) are still on their own round trip; branches on those values come afterwards.
    // [round 3] ORDER OF ISSUE = ORDER OF FIRST USE.  Loads return in the order they were issued and the CU's memory
    // pipeline takes 23 (128-bit) to 39 (32-bit) ticks per instruction (tools/ubench/cu_share.hip), so the thirty loads
    // of this prologue arrive over ~700 ticks behind the first: what forward kinematics needs -- joint positions,
    // base rows, index words, axis and lambda_H_pre -- goes first, the contact tables and inertias stream in behind it.
    // The joint rows are loaded BY LANE INDEX (lane l >= 1: row l - 1; an address that needs no table) and move to the
    // lane that owns the joint by one shuffle when the index word has arrived -- none when the joints are numbered in
    // lane order (P.jrow_seq) -- instead of being issued only then (a second round trip: 400 cycles).
    const VI jl = vsel(lane < 1, lane * 0, vsel(lane <= P.n, lane - 1, lane * 0 + (P.n - 1)));
    V s = V(T(0)), sd = V(T(0));
    if (kLaneRows) {
      s = ln.gload(A.state_in, jl + P.row_s, P.n_rows);
      sd = ln.gload(A.state_in, jl + P.row_sd, P.n_rows);
    }
    V pB[3], q[4], vW[3], om[3];
    // The 13 environment-uniform rows (base position, quaternion, base velocity): one load instruction per row
    // costs the CU's vector-memory pipeline 32 ticks each (tools/ubench/cu_share.hip; the prologue is bound by
    // that pipeline, which the waves of a CU share).  With an LDS area at hand lane k < 13 loads row k -- ONE
    // instruction -- and the values reach every lane through one LDS write and four broadcast reads.
    const bool stage_base = G >= 16 && A.has_lds != 0;
    V base_stage = V(T(0));
    if (stage_base) {
      const VI brow = vsel(lane < 7, lane, vsel(lane < 13, lane + P.n, lane * 0));  // rows 0..6 and 7+n..12+n
      base_stage = ln.gload(A.state_in, brow, P.n_rows);
    } else {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      pB[k] = ln.gload_u(A.state_in, P.row_pos + k, P.n_rows);
      vW[k] = ln.gload_u(A.state_in, P.row_vlin + k, P.n_rows);
      om[k] = ln.gload_u(A.state_in, P.row_vang + k, P.n_rows);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) q[k] = ln.gload_u(A.state_in, P.row_quat + k, P.n_rows);
    }
    const VI jtype = ln.lconsti(A.lti, LI_JTYPE);
    const VI parent = ln.lconsti(A.lti, LI_PARENT);
    const VI level = ln.lconsti(A.lti, LI_LEVEL);
    const VI lnk = ln.lconsti(A.lti, LI_LINK);   // reference link index of this lane
    const VI jrow = ln.lconsti(A.lti, LI_JROW);  // joint arrays are 0-based: ii = i - 1 (rbda/aba.py:133)
    VI jump[kMaxRounds], child[kMaxChildren];
#pragma unroll
    for (int k = 0; k < kMaxRounds; ++k) jump[k] = ln.lconsti(A.lti, LI_JUMP + k);  // -1 beyond the last round
#pragma unroll
    for (int k = 0; k < kMaxChildren; ++k) child[k] = ln.lconsti(A.lti, LI_CHILD + k);
    V ax[3], Rpre[9], ppre[3], cL[3], IL[6];
#pragma unroll
    for (int k = 0; k < 3; ++k) ax[k] = ln.lconstf(A.ltf, LF_AXIS + k);
#pragma unroll
    for (int k = 0; k < 9; ++k) Rpre[k] = ln.lconstf(A.ltf, LF_RPRE + k);
#pragma unroll
    for (int k = 0; k < 3; ++k) ppre[k] = ln.lconstf(A.ltf, LF_PPRE + k);
    const V mass = ln.lconstf(A.ltf, LF_MASS);
    V Rsuc[9], psuc[3];  // identity for most models (P.any_suc says whether they are used): three wide loads
#pragma unroll
    for (int k = 0; k < 9; ++k) Rsuc[k] = ln.lconstf(A.ltf, LF_RSUC + k);
#pragma unroll
    for (int k = 0; k < 3; ++k) psuc[k] = ln.lconstf(A.ltf, LF_PSUC + k);
    if (kLaneRows) ln.fence();  // ---- everything above is in flight before anything below is issued ----
    // collidable-point tables of chunk 0 (further chunks are loaded inside the contact loop); the tables
    // hold at least one (empty) slot per lane, so the load is legal whatever the contact model
    PointSlot ps0;
    if (kStep) load_slot_tables(lane, 0, ps0);
    // deformation rows of the points: by lane index too (lane l: point row l) when all collidable points fit one chunk
    const bool m_by_lane = kStep && !kRigid && kLaneRows && P.n_chunks == 1 && P.n_points <= G;
    V m_raw[3] = {V(T(0)), V(T(0)), V(T(0))};
    if (m_by_lane) {
      const VI pl = vsel(lane < P.n_points, lane, lane * 0) * 3;
#pragma unroll
      for (int k = 0; k < 3; ++k) m_raw[k] = ln.gload(A.state_in, pl + (P.row_m + k), P.n_rows);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) cL[k] = ln.lconstf(A.ltf, LF_COM + k);
#pragma unroll
    for (int k = 0; k < 6; ++k) IL[k] = ln.lconstf(A.ltf, LF_ICOM + k);
    V smin, smax, klim, dlim, kfc, kfv;
    if (kStep) {
      smin = ln.lconstf(A.ltf, LF_SMIN), smax = ln.lconstf(A.ltf, LF_SMAX);
      klim = ln.lconstf(A.ltf, LF_KLIM), dlim = ln.lconstf(A.ltf, LF_DLIM);
      kfc = ln.lconstf(A.ltf, LF_KC), kfv = ln.lconstf(A.ltf, LF_KV);
    }
    // tables of the row-distributed ABA passes: always present in the model block, loaded whatever the layout
    // the model ends up using (the flag that decides lives in the parameter block, one scalar load away)
    RowTabs rt;
    if ((kStep || MODE == MODE_FD) && !kRigid) load_row_tabs(rt);
    const VI jrow_c = vsel(jrow >= 0, jrow, lane * 0);
    const VI jsrc = jrow_c + 1;  // the lane that loaded this lane's joint row
    if (!kLaneRows) {  // the joint rows need the lane's joint index, one dependent round trip behind the tables
      s = ln.gload(A.state_in, jrow_c + P.row_s, P.n_rows);
      sd = ln.gload(A.state_in, jrow_c + P.row_sd, P.n_rows);
    }
    ln.fence();
    if (kLaneRows && !P.jrow_seq) s = ln.shfl(s, jsrc), sd = ln.shfl(sd, jsrc);
    if (stage_base) {
      ln.lds_write(lane + kArea, base_stage, lane < 16);
      ln.lds_sync();
      V b16[16];
      ln.template lds_readv<16>(lane * 0 + kArea, b16);
      ln.lds_sync();
#pragma unroll
      for (int k = 0; k < 3; ++k) pB[k] = b16[k], vW[k] = b16[7 + k], om[k] = b16[10 + k];
#pragma unroll
      for (int k = 0; k < 4; ++k) q[k] = b16[3 + k];
    }
    ln.stamp_after(A, 20, jrow);   // profiling build: index tables arrived
    ln.stamp_after(A, 21, q[3]);   // base rows of the state arrived
    ln.stamp_after(A, 22, ps0.hd); // point-slot tables arrived
    ln.stamp_after(A, 23, sd);     // joint rows of the state arrived (dependent on the index table)
    // Stage B: loads that need the rest of the argument block.
    if (MODE == MODE_ID && A.id_zero_vel) {
      sd = V(T(0));
#pragma unroll
      for (int k = 0; k < 3; ++k) vW[k] = V(T(0)), om[k] = V(T(0));
    }
    // The optional inputs.  [round 3] No load sits in a branch up here: a load issued on one side of a branch leaves
    // its destination register "pending" at the join, and the wait-count bookkeeping of the compiler then makes the
    // first instruction that reuses the register wait for EVERY load of the prologue -- forward kinematics started
    // when the last table had arrived instead of the ninth load.  The joint torques are loaded unconditionally (from
    // the joint-position rows when there are none: a row that exists; the value is dropped), last in the queue, and
    // first used by the actuation model right before ABA; the link wrenches are loaded where they are used.
    const bool has_tau = A.tau != nullptr;
    // [round 4] controlled rollout (jxs_rollout_controlled, KArgs::flags bit 2): `tau` holds one block of joint torques
    // per fused step, [n_steps * n][N]; step `it` reads rows it * n ... (reloaded at the top of every step below)
    const bool tau_seq = MODE == MODE_ROLLOUT && has_tau && (A.flags & 4) != 0;
    const int tau_rows = tau_seq ? A.n_steps * P.n : P.n;
    V tau_in;
    if (kLaneRows) {
      tau_in = ln.gload(has_tau ? A.tau : A.state_in, jl + (has_tau ? 0 : P.row_s), has_tau ? tau_rows : P.n_rows);
    } else {
      tau_in = has_tau ? ln.gload(A.tau, jrow_c, tau_rows) : V(T(0));
    }
    const bool with_rows = P.row_mode && (kStep || MODE == MODE_FD) && !kRigid;
    const bool with_contacts = (kStep && !kRigid) && P.n_chunks > 0;  // soft contacts (state m)

    const VM is_joint = jtype != 0;
    const VM is_rev = jtype == 1;
    const VM is_pri = jtype == 2;
    const VM is_root = level == 0;
    s = vsel(is_joint, s, V(T(0)));
    sd = vsel(is_joint, sd, V(T(0)));
    // Batch 2: the tangential deformation rows.  Loaded by lane index: they move to the lanes of their slots where they
    // are first used (the contact phase; RungeKutta4 keeps a copy of the initial state and needs them here).  Else
    // their addresses depend on the slot table just loaded.
    auto place_m = [&]() {
      const VI psrc = vsel(ps0.body >= 0, ps0.prow, lane * 0);
#pragma unroll
      for (int k = 0; k < 3; ++k) ps0.m[k] = P.prow_seq ? m_raw[k] : ln.shfl(m_raw[k], psrc);
    };
    if (with_contacts) {
      if (!m_by_lane) load_slot_state(ps0);
      else if (kRK4) place_m();
    }
    ln.stamp(A, 1);  // tables + state arrived

    // Fused rollout: `n_steps` consecutive steps with the state carried in registers -- tables,
    // inputs and state are read once, the state is written once (jxs_rollout).  One step otherwise.
    // ---- the state block of this environment: written back once -- and, for a recorded rollout, after every step ----
    auto write_state = [&](T* dst, int roff, int nrows) {
      const VI zl = lane * 0;
      ln.gstore(dst, roff + jrow + P.row_s, s, is_joint, nrows);
      ln.gstore(dst, roff + jrow + P.row_sd, sd, is_joint, nrows);
      if (G >= 16) {
        // [round 3] the thirteen base rows with ONE store instruction: every lane of an environment carries the same
        // new base state (it was integrated from values that are uniform over the environment), so lane k < 13 picks
        // value k and stores row k (rows 0..6 and 7+n..12+n) -- instead of thirteen exec-masked stores of the root lane
        const V b13[13] = {pB[0], pB[1], pB[2], q[0], q[1], q[2], q[3], vW[0], vW[1], vW[2], om[0], om[1], om[2]};
        // (a tree over the bits of the lane index: four masks and a depth of four, where a chain of twelve
        // compare-and-select pairs cost a wait state each -- a compare writes the mask its select reads)
        const VM b0 = (lane & 1) != 0, b1 = (lane & 2) != 0, b2 = (lane & 4) != 0, b3 = (lane & 8) != 0;
        const V t01 = vsel(b0, b13[1], b13[0]), t23 = vsel(b0, b13[3], b13[2]), t45 = vsel(b0, b13[5], b13[4]);
        const V t67 = vsel(b0, b13[7], b13[6]), t89 = vsel(b0, b13[9], b13[8]), tab = vsel(b0, b13[11], b13[10]);
        const V q0 = vsel(b1, t23, t01), q1 = vsel(b1, t67, t45), q2 = vsel(b1, tab, t89);
        const V val = vsel(b3, vsel(b2, b13[12], q2), vsel(b2, q1, q0));
        const VI brow = vsel(lane < 7, lane, vsel(lane < 13, lane + P.n, lane * 0));
        ln.gstore(dst, roff + brow, val, lane < 13, nrows);
      } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) ln.gstore(dst, roff + zl + (P.row_quat + k), q[k], is_root, nrows);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        ln.gstore(dst, roff + zl + (P.row_pos + k), pB[k], is_root, nrows);
        ln.gstore(dst, roff + zl + (P.row_vlin + k), vW[k], is_root, nrows);
        ln.gstore(dst, roff + zl + (P.row_vang + k), om[k], is_root, nrows);
      }
      }
      if (with_contacts) {
        const VM valid = ps0.body >= 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) ln.gstore(dst, roff + ps0.prow * 3 + (P.row_m + k), ps0.m[k], valid, nrows);
      }
    };
    const int n_steps = (MODE == MODE_ROLLOUT) ? A.n_steps : 1;  // compile-time 1 for a plain step
    for (int it = 0; it < n_steps; ++it) {
    V tau = V(T(0));  // (set at stage 0, right before ABA: see "B: joint torques" below)
    if (MODE == MODE_ROLLOUT && tau_seq && it > 0)  // this step's torques: issued here, first used right before ABA
      tau_in = kLaneRows ? ln.gload(A.tau, jl + it * P.n, tau_rows) : ln.gload(A.tau, jrow_c + it * P.n, tau_rows);

    // Runge-Kutta 4 (api/integrators.py:91-167): the joint torques (B, below) are computed once from the
    // initial state (api/model.py:2658), the stage loop below evaluates system_dynamics
    // (api/ode.py:174-225) at x0, x0 + dt/2 k1, x0 + dt/2 k2, x0 + dt k3.  One stage for Euler.
    // RigidContacts: stage 0 is the step with QP contact forces, stage 1 re-evaluates the kinematics
    // and the articulated inertias at the new state for the impact (rbda/contacts/rigid.py:391-446).
    // RigidContacts appends one more pass over the new state for the impact (kImpactStage).
    constexpr int kImpactStage = kRK4 ? 4 : 1;
    constexpr int n_stages = kDyn ? 1 : (kRigid ? kImpactStage + 1 : (kRK4 ? 4 : 1));
    V x0s, x0sd, x0q[4], x0p[3], x0v[3], x0w[3], x0m[3];  // stage-0 state (quaternion normalised)
    V ks, ksd, kq[4], kp[3], kv[3], kw[3], km[3];           // weighted sum of the stage derivatives
    V xfl[3], xfa[3];                                       // external link wrench in the stage-0 frame C
    // RungeKutta4Fast (api/integrators.py:170-276): contact link wrench and position derivatives of stage 0
    V xcfl[3], xcfa[3], sd0, dq0[4], pd0[3];
    if (kRK4) {
      const V nrm0 = vsqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
      const V inv0 = vrcp(vsel(nrm0 == V(T(0)), V(T(1)), nrm0));
      x0s = s, x0sd = sd, ks = V(T(0)), ksd = V(T(0));
#pragma unroll
      for (int k = 0; k < 4; ++k) q[k] = q[k] * inv0, x0q[k] = q[k], kq[k] = V(T(0));
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        x0p[k] = pB[k], x0v[k] = vW[k], x0w[k] = om[k], x0m[k] = ps0.m[k];
        kp[k] = V(T(0)), kv[k] = V(T(0)), kw[k] = V(T(0)), km[k] = V(T(0));
      }
    }
    // (the RungeKutta4 + rigid-contact kernel keeps this loop rolled: one copy of the contact solvers)
#pragma unroll(kRK4 && kRigid ? 1 : 8)
    for (int stage = 0; stage < n_stages; ++stage) {
    if (kRigid && stage == kImpactStage && P.rigid == 2) break;  // RelaxedRigidContacts: no velocity reset (relaxed_rigid.py:265-281)
    // ---- base rotation: DCM of q/|q| (data.base_orientation, api/data.py:267-286) --------
    V R[9], r[3];
    V R0[9];
    base_dcm(q, R0);
    V fkrec[kDuoFkRec];  // two-wave workgroups: link pose and anchors as published by the inertia wave
    if constexpr (ROLE == ROLE_MAIN) {
      ln.stamp(A, 2);
      ln.flag_wait(kDuoFlagFk, duo_seen_);
      ln.template lds_readv<kDuoFkRec>(lane * kDuoFkRec + duo_fk_off(G), fkrec);
#pragma unroll
      for (int e = 0; e < 9; ++e) R[e] = fkrec[e];
#pragma unroll
      for (int e = 0; e < 3; ++e) r[e] = fkrec[9 + e];
      ln.stamp_after(A, 3, r[2]);
    } else {
    // ---- H: local parent->child transform lambda_H_pre * pre_H_suc(s) * suc_H_i ----------
    //      (api/kin_dyn_parameters.py:396-451, math/joint_model.py:146-200, math/rotation.py:58-84)
    local_transform(is_rev, is_pri, is_root, s, ax, Rpre, ppre, Rsuc, psuc, R0, R, r);
    ln.stamp(A, 2);  // actuation + local transforms

    // ---- M: forward kinematics as a tree prefix product (pointer jumping) ----------------
    //      equals the scan of rbda/forward_kinematics.py:80-103 in exact arithmetic
    fk_prefix(jump, R, r);
    ln.stamp(A, 3);  // forward kinematics
    }

    // Anchored ABA (see "G: articulated-body algorithm" below): origin of the leaf link of this lane's
    // first-child chain, and of the parent's chain -- fetched here so that the shuffles complete behind the
    // velocity and contact phases.
    constexpr bool kAnchMode = (kStep && !kRigid) || MODE == MODE_FD;
    const bool anch = kAnchMode && P.anchored != 0;
    V ra[3] = {V(T(0)), V(T(0)), V(T(0))}, dpl[3] = {V(T(0)), V(T(0)), V(T(0))};
    if constexpr (ROLE == ROLE_MAIN) {
#pragma unroll
      for (int e = 0; e < 3; ++e) ra[e] = fkrec[12 + e], dpl[e] = fkrec[15 + e];  // (zeros without anchors)
    } else if (anch) {
      const VI anchor = ln.lconsti(A.lti, LI_ANCHOR), panchor = ln.lconsti(A.lti, LI_PANCHOR);
      V rq[3];
#pragma unroll
      for (int e = 0; e < 3; ++e) ra[e] = ln.shfl(r[e], anchor), rq[e] = ln.shfl(r[e], panchor);
#pragma unroll
      for (int e = 0; e < 3; ++e) dpl[e] = ra[e] - rq[e];
    }

    // Base velocity in C: [v_W + w x p_B ; w] (= mixed velocity).  ABA keeps v_0 = 0 for a
    // fixed base (rbda/aba.py:109-121) while the cached link velocities still start from the
    // stored base velocity (rbda/forward_kinematics.py:69-70): `vBx` carries that difference.
    V vBc[3];
    cross(om, pB, vBc);
#pragma unroll
    for (int k = 0; k < 3; ++k) vBc[k] = vBc[k] + vW[k];

    // ---- motion subspace in C: S = W_X_i [0;a] or W_X_i [a;0]  (kin_dyn_parameters.py:239-261)
    V Sl[3], Sa[3], Ra_[3];
    mat3vec(R, ax, Ra_);
    {
      V rxa[3];
      cross(r, Ra_, rxa);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        if (P.any_pri) {
          Sa[k] = vsel(is_rev, Ra_[k], V(T(0)));
          Sl[k] = vsel(is_rev, rxa[k], vsel(is_pri, Ra_[k], V(T(0))));
        } else {  // every joint is revolute, and the lanes without a joint carry a zero axis in the table: nothing to mask
          Sa[k] = Ra_[k];
          Sl[k] = rxa[k];
        }
      }
    }

    // ---- link velocities: v_i = v_lambda + S_i sd_i  -> prefix sum over ancestors --------
    V vJl[3], vJa[3], vl[3], va[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      vJl[k] = Sl[k] * sd;
      vJa[k] = Sa[k] * sd;
      vl[k] = vsel(is_root, P.floating ? vBc[k] : V(T(0)), vJl[k]);
      va[k] = vsel(is_root, P.floating ? om[k] : V(T(0)), vJa[k]);
    }
    prefix6(jump, vl, va);

    // c_i = v_i x vJ_i   (Cross.vx, math/cross.py:14-43; rbda/aba.py:141)
    V cl[3], ca[3];
    {
      V t0[3], t1[3];
      cross(va, vJl, t0);
      cross(vl, vJa, t1);
      cross(va, vJa, ca);
#pragma unroll
      for (int k = 0; k < 3; ++k) cl[k] = t0[k] + t1[k];
    }
    ln.stamp(A, 4);  // velocities

    // offset of the cached kinematics w.r.t. the ABA base frame (quirk 12): R0 * p(suc_H_i[0])
    V doff[3];
    {
      V bo[3] = {V(P.base_off[0]), V(P.base_off[1]), V(P.base_off[2])};
      if (P.has_base_off) {
        mat3vec(R0, bo, doff);
      } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) doff[k] = V(T(0));
      }
    }

    if (MODE == MODE_KIN) {
      store_kinematics(lnk, level, R, r, vl, va, pB, doff, vBc, om);
      return;
    }
    if (MODE == MODE_JAC) {
      jacobians(lane, lnk, level, jrow, is_joint, is_root, R0, R, r, Sl, Sa, vl, va, vBc, om);
      return;
    }
    if (MODE == MODE_GRAV) {
      gravity_torques(lane, jrow, level, child, is_joint, R, r, cL, mass, Sl, Sa);
      return;
    }

    // ---- external link wrenches in C -----------------------------------------------------
    V fl[3] = {V(T(0)), V(T(0)), V(T(0))}, fa[3] = {V(T(0)), V(T(0)), V(T(0))};
    if (kRK4 && stage > 0) {
      // The inertial wrench is converted once, with the link transforms of the initial state
      // (api/model.py:2641-2646), and held over the stages: only its moment is re-referred to the
      // origin of this stage's frame C,  mu_Cs = mu_C0 + (p_B0 - p_Bs) x f.
      if (A.link_f != nullptr) {
        V dp[3] = {x0p[0] - pB[0], x0p[1] - pB[1], x0p[2] - pB[2]}, t[3];
        cross(dp, xfl, t);
#pragma unroll
        for (int k = 0; k < 3; ++k) fl[k] = xfl[k], fa[k] = xfa[k] + t[k];
      }
    } else if (A.link_f != nullptr) {
      const VM is_link = level >= 0;
      V f6[6];
      {
        const VI lrow = vsel(lnk >= 0, lnk, lane * 0) * 6;
#pragma unroll
        for (int k = 0; k < 6; ++k) f6[k] = ln.gload(A.link_f, lrow + k, P.nL * 6);
#pragma unroll
        for (int k = 0; k < 6; ++k) f6[k] = vsel(is_link, f6[k], V(T(0)));
      }
      V arm[3], t[3];
      if (A.force_repr == REPR_INERTIAL) {
        // [f; mu_W - p_B x f]
        cross(pB, f6, t);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          fl[k] = f6[k];
          fa[k] = f6[3 + k] - t[k];
        }
      } else {
        // Mixed: [f; mu + p_L x f]; Body: [R f; R mu + p_L x R f]  (api/common.py:191-219),
        // p_L from the cached link transforms (api/model.py:2641-2646).
        V fw[3], mw[3];
        if (A.force_repr == REPR_BODY) {
          mat3vec(R, f6, fw);
          mat3vec(R, f6 + 3, mw);
        } else {
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            fw[k] = f6[k];
            mw[k] = f6[3 + k];
          }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) arm[k] = r[k] + doff[k];
        cross(arm, fw, t);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          fl[k] = fw[k];
          fa[k] = mw[k] + t[k];
        }
      }
      if (kRK4) {
#pragma unroll
        for (int k = 0; k < 3; ++k) xfl[k] = fl[k], xfa[k] = fa[k];
      }
    }

    // anchored ABA: from here on the link wrench sums (fl, fa) are referred to the lane's anchor -- the
    // external wrench is shifted once, the contact moments below are formed about the anchor directly
    // (lever = centimetres instead of the distance to the base: (r_C - ra) x f instead of r_C x f - ra x f)
    if (anch && A.link_f != nullptr) {
      V t[3];
      cross(ra, fl, t);
#pragma unroll
      for (int k = 0; k < 3; ++k) fa[k] = fa[k] - t[k];
    }

    // ---- J,K,L,I: soft contacts ----------------------------------------------------------
    // MODE_DYN*: the contact wrench of every link is an output (api/contact.py:514-603): summed on its own (about the
    // link's anchor, in C), then added to the external wrench
    V dcfl[3] = {V(T(0)), V(T(0)), V(T(0))}, dcfa[3] = {V(T(0)), V(T(0)), V(T(0))};
    V* const cacc_l = kDyn ? dcfl : fl;
    V* const cacc_a = kDyn ? dcfa : fa;
    // inertial wrench [f; mu_C + (ra + p_B) x f] of the link of this lane, rows 6 * link ... of KArgs::out_H
    auto store_link_contact_wrenches = [&]() {
      V arm[3] = {ra[0] + pB[0], ra[1] + pB[1], ra[2] + pB[2]}, t[3];
      cross(arm, dcfl, t);
      const VM is_link = level >= 0;
      const VI lrow = vsel(lnk >= 0, lnk, lane * 0) * 6;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        ln.gstore(A.out_H, lrow + k, dcfl[k], is_link, P.nL * 6);
        ln.gstore(A.out_H, lrow + (3 + k), dcfa[k] + t[k], is_link, P.nL * 6);
      }
    };
    if (with_contacts && m_by_lane && !kRK4 && it == 0 && stage == 0) place_m();
    if (with_contacts && with_rows && P.n_chunks == 1) {
      // The kinematics of the parent links reach the point lanes through the LDS scratch of the
      // row-distributed layout (18 writes + 18 reads, one round trip) instead of 18 ds_bpermute:
      // measured 9.70 -> 9.56 us per step (ds_bpermute issues every ~22 cycles for a lone wave).
      const int KIN = lds_kin_offset(G) + kArea;
      const VM valid = ps0.body >= 0;
      V m[3], Rb[9], rb[3], vbl[3], vba[3], rab[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) m[k] = vsel(valid, ps0.m[k], V(T(0)));
      const VI body_c = vsel(valid, ps0.body, lane * 0);
      if constexpr (ROLE == ROLE_MAIN) {
        // pose and anchor of the parent link: straight from the record the inertia wave published (wide reads);
        // only the link velocities are staged here (8 words per link)
        const V v8[8] = {vl[0], vl[1], vl[2], va[0], va[1], va[2], V(T(0)), V(T(0))};
        ln.template lds_writev<8>(lane * 8 + KIN, v8);
        ln.lds_sync();
        V pr16[16], pv8[8];
        ln.template lds_readv<16>(body_c * kDuoFkRec + duo_fk_off(G), pr16);
        ln.template lds_readv<8>(body_c * 8 + KIN, pv8);
#pragma unroll
        for (int e = 0; e < 9; ++e) Rb[e] = pr16[e];
#pragma unroll
        for (int e = 0; e < 3; ++e) rb[e] = pr16[9 + e], rab[e] = pr16[12 + e], vbl[e] = pv8[e], vba[e] = pv8[3 + e];
        ln.lds_sync();
      } else {
      // one record of kKinRec = 24 words per link, written and read with 128-bit LDS instructions (an LDS
      // instruction costs a wave 11..15 ticks of issue whatever its width, tools/ubench/cu_share.hip: six writes
      // and six reads instead of twenty-one of each)
      const V k24[kKinRec] = {R[0], R[1], R[2], R[3], R[4], R[5], R[6], R[7], R[8], r[0], r[1], r[2], vl[0], vl[1], vl[2],
                              va[0], va[1], va[2], ra[0], ra[1], ra[2], V(T(0)), V(T(0)), V(T(0))};
      ln.template lds_writev<kKinRec>(lane * kKinRec + KIN, k24);
      ln.lds_sync();
      V p24[kKinRec];
      ln.template lds_readv<kKinRec>(body_c * kKinRec + KIN, p24);
#pragma unroll
      for (int e = 0; e < 9; ++e) Rb[e] = p24[e];
#pragma unroll
      for (int e = 0; e < 3; ++e) rb[e] = p24[9 + e], vbl[e] = p24[12 + e], vba[e] = p24[15 + e], rab[e] = p24[18 + e];
      ln.lds_sync();
      }
      V w6[6], mdl[3];
      point_physics(valid, ps0.Lp, m, Rb, rb, vbl, vba, rab, pB, doff, vBc, om, w6, mdl);
#pragma unroll
      for (int k = 0; k < 3; ++k) ps0.md[k] = mdl[k];
      link_wrench_sums(lane, ps0.tail, ps0.hd, w6, cacc_l, cacc_a);
    } else
    if (with_contacts) contacts<kRK4, kDyn>(lane, ps0, R, r, vl, va, ra, pB, doff, vBc, om, cacc_l, cacc_a, stage);  // sets ps0.md
    if constexpr (kDyn && !kRigid) {
#pragma unroll
      for (int k = 0; k < 3; ++k) fl[k] = fl[k] + dcfl[k], fa[k] = fa[k] + dcfa[k];
      if (A.out_H != nullptr) store_link_contact_wrenches();
      if (A.state_out == nullptr) return;  // js.contact.link_contact_forces alone: the accelerations are not asked for
    }
    ln.stamp(A, 5);  // contacts

    // ---- B: joint torques (api/actuation_model.py:7-126): once per step, from the state the step starts from (stage 0
    //      of RungeKutta4, api/model.py:2658).  Placed here, not at the top: the torque input is the LAST load of the
    //      prologue, and nothing before ABA needs it.
    if (stage == 0) {
      V tq = tau_in;
      if (kLaneRows) {
        if (!P.jrow_seq) tq = ln.shfl(tq, jsrc);
        tq = has_tau ? tq : V(T(0));
      }
      tau = vsel(is_joint, tq, V(T(0)));
      if constexpr (kRigid) {
        // [round 4] gravity-compensated step (jxs_step_gravity_compensated): tau_ref += g(q), the joint part of
        // free_floating_gravity_forces (api/model.py:1897-1931) of the state the step starts from -- the controller
        // loop "tau = g(q); step(tau)" of BASELINE config 5 in ONE launch (the kinematics are in registers anyway)
        if (A.flags & 2) tau = tau + vsel(is_joint, gravity_tq(level, child, R, r, cL, mass, Sl, Sa), V(T(0)));
      }
      if (kStep && !kDyn) {  // (system_dynamics takes the joint torques as they are: api/ode.py:117-122)
        const V lower = vmin(s - smin, V(T(0)));  // clip(max=0)
        const V upper = vmax(s - smax, V(T(0)));  // clip(min=0)
        V tau_pl = -(klim * (lower + upper));
        // `jnp.positive` is unary plus: tau_pl -= tau_pl * d * sd   (actuation_model.py:64-66)
        tau_pl = tau_pl - tau_pl * (dlim * sd);
        V tau_fr = V(T(0));
        if (P.enable_friction) {
          const V sgn = vsel(sd > V(T(0)), V(T(1)), vsel(sd < V(T(0)), V(T(-1)), V(T(0))));
          tau_fr = -(kfc * sgn + kfv * sd);
        }
        const V tot = tau + tau_fr + tau_pl;
        const V av = vabs(sd);
        const V lim = vsel(av <= V(P.w_th), V(P.tau_max),
                           vsel(av <= V(P.w_max), P.tau_max * (V(T(1)) - (av - P.w_th) * P.inv_w_range), V(T(0))));
        tau = vmax(vmin(tot, lim), -lim);  // clip(tot, -lim, lim)
      }
    }

    // ---- link inertia in C and bias force --------------------------------------------------
    // M = [[m I, m S(c)^T],[m S(c), I_c + m S(c) S(c)^T]]  (math/inertia.py:14-41) with the CoM
    // c = r + R c_L and I_c = R I_L R^T expressed in C.
    V cw[3], Ic[6];
    link_inertia_C(R, r, cL, IL, cw, Ic);
    // h = M v:  h_lin = m (v + w x c),  h_ang = I_c w + c x h_lin
    V hl[3], ha[3];
    {
      V t[3];
      cross(va, cw, t);
#pragma unroll
      for (int k = 0; k < 3; ++k) hl[k] = mass * (vl[k] + t[k]);
      V Iw[3];
      Iw[0] = Ic[0] * va[0] + Ic[1] * va[1] + Ic[2] * va[2];
      Iw[1] = Ic[1] * va[0] + Ic[3] * va[1] + Ic[4] * va[2];
      Iw[2] = Ic[2] * va[0] + Ic[4] * va[1] + Ic[5] * va[2];
      cross(cw, hl, t);
#pragma unroll
      for (int k = 0; k < 3; ++k) ha[k] = Iw[k] + t[k];
    }
    // v x* h = [w x h_lin ; v x h_lin + w x h_ang]   (Cross.vx_star, math/cross.py:45-58)
    V bl[3], ba[3];
    {
      V t0[3], t1[3];
      cross(va, hl, bl);
      cross(vl, hl, t0);
      cross(va, ha, t1);
#pragma unroll
      for (int k = 0; k < 3; ++k) ba[k] = t0[k] + t1[k];
    }

    if (MODE == MODE_ID) {
      rnea(lane, jrow, level, jump, child, is_joint, is_root, Sl, Sa, cl, ca, mass, cw, Ic, bl, ba, fl, fa, pB);
      return;
    }

    // ---- G: articulated-body algorithm ------------------------------------------------------
    // Anchored ABA.  With one reference point for the whole tree (the origin of C) the articulated inertias
    // of the light distal links carry parallel-axis terms m |r|^2 of a lever r ~ 1 m that must cancel in
    // d = S^T MA S; in fp32 that costs a median step error of 8e-6 where the reference formulation (link
    // coordinates) has 1e-7 (tools/fp32_error.py).  Moving the origin of C to the feet helps the ankles and hurts
    // the arms by the same factor (measured), so every FIRST-CHILD CHAIN gets its own reference point: the
    // origin of its leaf link (`ra`, relative to the origin of C).  Along a chain nothing changes -- parents
    // still simply add -- and only where a non-first child joins its parent (a branching link) the inertia
    // is re-referred by `dpl` = anchor(child chain) - anchor(parent chain) (xlate_inertia).  Motion vectors
    // shift as [lin - ra x ang; ang], force vectors as [f; n - ra x f].
    if (anch) {
      V t[3];
      cross(ra, Sa, t);
#pragma unroll
      for (int k = 0; k < 3; ++k) Sl[k] = Sl[k] - t[k];
      cross(ra, ca, t);
#pragma unroll
      for (int k = 0; k < 3; ++k) cl[k] = cl[k] - t[k], cw[k] = cw[k] - ra[k];
    }
    // pA_i = v x* M v - f_i          (rbda/aba.py:109-121,157-160)
    V pA[6];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      pA[k] = bl[k] - fl[k];
      pA[3 + k] = ba[k] - fa[k];
    }
    if (anch) {  // the bias force moves to the anchor; (fl, fa) are referred to it already
      V t[3];
      cross(ra, bl, t);
#pragma unroll
      for (int k = 0; k < 3; ++k) pA[3 + k] = pA[3 + k] - t[k];
    }
    // MA_i = M_i (upper triangle of the symmetric 6x6)
    V MA[21];
    assemble_MA(mass, cw, Ic, MA);
    const V S6[6] = {Sl[0], Sl[1], Sl[2], Sa[0], Sa[1], Sa[2]};
    const V c6[6] = {cl[0], cl[1], cl[2], ca[0], ca[1], ca[2]};
    ln.stamp(A, 6);  // inertia + bias

    if (MODE == MODE_CRBA) {
      crba(lane, level, child, jrow, is_joint, is_root, MA, S6);
      return;
    }

    V sdd = V(T(0));
    V acl[3], aca[3];  // base spatial acceleration in C incl. gravity (valid in every lane)
    if (P.row_mode && (kStep || MODE == MODE_FD) && !kRigid) {
      V a0[6];
      if constexpr (ROLE == ROLE_MAIN)
        aba_rows_main(lane, rt, pA, S6, c6, tau, anch, dpl, sdd, a0);
      else
        aba_rows(lane, rt, MA, pA, S6, c6, tau, anch, dpl, sdd, a0);
      if (anch && P.floating) {  // from the base chain's anchor back to the origin of C: a_lin - alpha x ra_0
        V ra0[3], t[3];
        const VI zero_lane = lane * 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) ra0[k] = ln.shfl(ra[k], zero_lane);
        ln.fence();
        cross(a0 + 3, ra0, t);
#pragma unroll
        for (int k = 0; k < 3; ++k) a0[k] = a0[k] - t[k];
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        acl[k] = P.floating ? a0[k] : V(T(0));
        aca[k] = P.floating ? a0[3 + k] : V(T(0));
      }
      if (P.floating) acl[2] = acl[2] + P.g;
    } else {
      // Pass 2 (rbda/aba.py:184-224), leaves to base, one tree level per iteration.  In frame C
      // the propagation X^T Ma X is the identity congruence: parents simply add.
      V U[6], inv_d = V(T(0)), u = V(T(0));
  #pragma unroll
      for (int k = 0; k < 6; ++k) U[k] = V(T(0));
      // fixed base: nothing propagates into link 0 -- except for the mass-matrix inverse, where the reference treats
      // the base of every model as a free body (rbda/mass_inverse.py:118-178)
      // ... and for the rigid contact models, whose contact solve uses the inverse of the FULL free-floating mass matrix
      // for every model (rbda/contacts/rigid.py:296, relaxed_rigid.py:370): the base factor below answers unit wrenches
      // as a free body there, while the dynamics themselves (pass 3, the response to the solved forces) keep it fixed
      const int first_level = (P.floating || MODE == MODE_MINV || kRigid) ? 1 : 2;
      const int max_depth = P.max_depth;
      const unsigned long long mc0 = P.maxch_nib[0], mc1 = P.maxch_nib[1], mc2 = P.maxch_nib[2], mc3 = P.maxch_nib[3];
      const unsigned long long nonadj = P.nonadj_levels;
      for (int Lv = max_depth; Lv >= 1; --Lv) {
        // U = MA S, d = S^T U, u = tau - S^T pA
  #pragma unroll
        for (int i = 0; i < 6; ++i) {
          V acc = MA[sidx(i, 0)] * S6[0];
  #pragma unroll
          for (int j = 1; j < 6; ++j) acc = acc + MA[sidx(i, j)] * S6[j];
          U[i] = acc;
        }
        V d = U[0] * S6[0], sp = pA[0] * S6[0];
  #pragma unroll
        for (int i = 1; i < 6; ++i) {
          d = d + U[i] * S6[i];
          sp = sp + pA[i] * S6[i];
        }
        u = tau - sp;
        inv_d = vsel(is_joint, vrcp_acc(d), V(T(0)));  // finite everywhere: base / padding lanes have d = 0
        if (Lv < first_level) break;
        // Ma = MA - U U^T / d ;  pa = pA + Ma c + U u / d
        V Ma[21], pa[6], Ud[6];
  #pragma unroll
        for (int i = 0; i < 6; ++i) Ud[i] = U[i] * inv_d;
  #pragma unroll
        for (int i = 0; i < 6; ++i)
  #pragma unroll
          for (int j = i; j < 6; ++j) Ma[sidx(i, j)] = MA[sidx(i, j)] - Ud[i] * U[j];
  #pragma unroll
        for (int i = 0; i < 6; ++i) {
          V acc = pA[i] + Ud[i] * u;
  #pragma unroll
          for (int j = 0; j < 6; ++j) acc = acc + Ma[sidx(i, j)] * c6[j];
          pa[i] = acc;
        }
        // parents at level Lv-1 gather from their children (all at level Lv).  In the depth-first
        // lane order the first child sits in lane+1: a DPP lane shift, no LDS round trip.
        const VM is_par = level == (Lv - 1);
        const unsigned long long mcw = Lv < 16 ? mc0 : Lv < 32 ? mc1 : Lv < 48 ? mc2 : mc3;
        const int nch = (int)((mcw >> ((Lv & 15) * 4)) & 15ull);
        if (nch >= 1) {
          const VM ok0 = is_par && (child[0] >= 0);
          // 27 values = 3 blocks of 9 fused "acc += value(lane+1) * ok"
          V acc9[9], src9[9];
          add_from_next<9>(lane, MA, Ma, ok0);
          add_from_next<9>(lane, MA + 9, Ma + 9, ok0);
  #pragma unroll
          for (int e = 0; e < 3; ++e) {
            acc9[e] = MA[18 + e];
            src9[e] = Ma[18 + e];
          }
  #pragma unroll
          for (int e = 0; e < 6; ++e) {
            acc9[3 + e] = pA[e];
            src9[3 + e] = pa[e];
          }
          add_from_next<9>(lane, acc9, src9, ok0);
  #pragma unroll
          for (int e = 0; e < 3; ++e) MA[18 + e] = acc9[e];
  #pragma unroll
          for (int e = 0; e < 6; ++e) pA[e] = acc9[3 + e];
        }
  #pragma unroll
        for (int k = 1; k < kMaxChildren; ++k) {
          if (k >= P.max_children) break;  // (the widest link of the MODEL: a constant of a model-specialised kernel -- no copies, no child registers beyond it; the level's own width is tested below)
          if (k < nch) {
            // 1.0 where this lane is a parent of the current level with a k-th child, else 0.0
            const V okf = vsel(is_par && (child[k] >= 0), V(T(1)), V(T(0)));
            const int sh = (L::kHasRowShl && A.spec_consts && !anch) ? P.child_shift(k) : 0;
            if (sh > 0) {
              // [round 3] every such child sits `sh` lanes up in its parent's 16-lane row: gathered by a DPP row shift
              // folded into the accumulation (27 instructions instead of 27 ds_bpermute, a wait and 27 multiply-adds)
              V acc9[9], src9[9];
              ln.template fmac_row_shl<9>(MA, Ma, okf, sh);
              ln.template fmac_row_shl<9>(MA + 9, Ma + 9, okf, sh);
#pragma unroll
              for (int e = 0; e < 3; ++e) acc9[e] = MA[18 + e], src9[e] = Ma[18 + e];
#pragma unroll
              for (int e = 0; e < 6; ++e) acc9[3 + e] = pA[e], src9[3 + e] = pa[e];
              ln.template fmac_row_shl<9>(acc9, src9, okf, sh);
#pragma unroll
              for (int e = 0; e < 3; ++e) MA[18 + e] = acc9[e];
#pragma unroll
              for (int e = 0; e < 6; ++e) pA[e] = acc9[3 + e];
              continue;
            }
            // issue all 27 shuffles back to back, wait once, then consume (Ma/pa are finite in
            // every lane, see inv_d above, so masking by multiplication is safe)
            V gM[21], gp[6], gd[3];
  #pragma unroll
            for (int e = 0; e < 21; ++e) gM[e] = ln.shfl(Ma[e], child[k]);
  #pragma unroll
            for (int e = 0; e < 6; ++e) gp[e] = ln.shfl(pa[e], child[k]);
            if (anch) {
  #pragma unroll
              for (int e = 0; e < 3; ++e) gd[e] = ln.shfl(dpl[e], child[k]);
            }
            ln.fence();
            if (anch) xlate_inertia(gM, gp, gd);  // the child's chain has its own reference point
  #pragma unroll
            for (int e = 0; e < 21; ++e) MA[e] = MA[e] + okf * gM[e];
  #pragma unroll
            for (int e = 0; e < 6; ++e) pA[e] = pA[e] + okf * gp[e];
          }
        }
      }
      ln.stamp(A, 7);  // pass 2

      if (MODE == MODE_MINV) {
        mass_inverse(lane, level, parent, child, jrow, is_joint, is_root, MA, U, S6, inv_d);
        return;
      }

      // Pass 3 (rbda/aba.py:240-267): base acceleration, then top-down.
      V a6[6];
      if (P.floating) {
        solve6(MA, pA, a6);  // a0 = solve(-MA_0, pA_0), meaningful in the base lane only
      } else {
  #pragma unroll
        for (int k = 0; k < 6; ++k) a6[k] = V(T(0));
        a6[2] = V(-P.g);  // a0 = -B_X_W W_g expressed in C
      }
      ln.stamp(A, 8);  // base solve
      const VM par_adjacent = parent == (lane - 1);
      for (int Lv = 1; Lv <= max_depth; ++Lv) {
        V ap[6];
  #pragma unroll
        for (int k = 0; k < 6; ++k) ap[k] = ln.from_prev(a6[k]);  // parent in lane-1 (first children)
        if ((nonadj >> Lv) & 1ull) {
          V aq[6];
  #pragma unroll
          for (int k = 0; k < 6; ++k) aq[k] = ln.shfl(a6[k], parent);
          ln.fence();
  #pragma unroll
          for (int k = 0; k < 6; ++k) ap[k] = vsel(par_adjacent, ap[k], aq[k]);
        }
        const VM act = level == Lv;
        if (anch) {  // a(own anchor) = a(parent's anchor) + alpha x dpl   (dpl = 0 along a chain)
          V t[3];
          cross(ap + 3, dpl, t);
  #pragma unroll
          for (int k = 0; k < 3; ++k) ap[k] = ap[k] + t[k];
        }
        V ai[6];
  #pragma unroll
        for (int k = 0; k < 6; ++k) ai[k] = ap[k] + c6[k];
        V ua = U[0] * ai[0];
  #pragma unroll
        for (int k = 1; k < 6; ++k) ua = ua + U[k] * ai[k];
        const V sdd_i = (u - ua) * inv_d;
        sdd = vsel(act, sdd_i, sdd);
  #pragma unroll
        for (int k = 0; k < 6; ++k) a6[k] = vsel(act, ai[k] + S6[k] * sdd_i, a6[k]);
      }

      // Base acceleration: W_a = a_0 + W_g (rbda/aba.py:284-292); the angular part is frame
      // independent, the linear part is shifted back to the world origin at the end.
      {
        const VI zero_lane = lane * 0;
  #pragma unroll
        for (int k = 0; k < 3; ++k) {
          acl[k] = P.floating ? ln.shfl(a6[k], zero_lane) : V(T(0));
          aca[k] = P.floating ? ln.shfl(a6[3 + k], zero_lane) : V(T(0));
        }
        if (anch && P.floating) {  // from the base chain's anchor back to the origin of C: a_lin - alpha x ra_0
          V ra0[3], t[3];
  #pragma unroll
          for (int k = 0; k < 3; ++k) ra0[k] = ln.shfl(ra[k], zero_lane);
          ln.fence();
          cross(aca, ra0, t);
  #pragma unroll
          for (int k = 0; k < 3; ++k) acl[k] = acl[k] - t[k];
        }
        if (P.floating) acl[2] = acl[2] + P.g;
      }
      if (kRigid && P.n_chunks > 0) {
        TreeFac tf;
  #pragma unroll
        for (int k = 0; k < 6; ++k) tf.U[k] = U[k], tf.S6[k] = S6[k];
        tf.inv_d = inv_d;
        ldl6_factor(MA, tf);  // (fixed-base models too: the contact solves treat the base as a free body, see pass 2)
        RigidPoints rp;
        rigid_points(ps0, R, r, vl, va, pB, vBc, om, rp);
        const VI zero_lane = lane * 0;
        if (stage < kImpactStage) {
          // contact forces, then nudot = nudot_free + M^-1 J^T f  (api/ode.py:57-131); with RungeKutta4 at
          // every stage, as system_dynamics is (api/ode.py:174-225)
          V cfl[3] = {V(T(0)), V(T(0)), V(T(0))}, cfa[3] = {V(T(0)), V(T(0)), V(T(0))};
          if (kRK4 && P.rk4fast && stage > 0) {
            // RungeKutta4Fast: the inertial contact wrench of the initial state is held over the stages
            // (integrators.py:175-187); only its moment is re-referred to this stage's frame C
            V dp[3] = {x0p[0] - pB[0], x0p[1] - pB[1], x0p[2] - pB[2]}, t[3];
            cross(dp, xcfl, t);
  #pragma unroll
            for (int k = 0; k < 3; ++k) cfl[k] = xcfl[k], cfa[k] = xcfa[k] + t[k];
          } else if (P.rigid == 2 && P.n_chunks > 1) {
            // [round 5] more collidable points than lanes: chunks of G points, solved in the tree (jxs_rigid.inc)
            relaxed_contact_forces_chunked(lane, level, parent, child, tf, R, r, vl, va, pB, vBc, om, a6, mass, cw, Ic, cfl, cfa);
          } else {
            V fpt[3];
            if (P.rigid == 2)
              relaxed_contact_forces(lane, level, parent, child, tf, ps0, rp, a6, mass, cw, Ic, fpt);
            else
              rigid_contact_forces(lane, level, parent, child, tf, ps0, rp, a6, mass, cw, Ic, fpt);
            V w6[6];
  #pragma unroll
            for (int k = 0; k < 3; ++k) w6[k] = fpt[k];
            cross(rp.rc, w6, w6 + 3);
            scatter_point_wrenches(lane, ps0, w6, cfl, cfa);
#ifdef JXS_RIGID_DEBUG
            for (int i = 0; i < G; ++i) if (cfl[2].v[i] != 0) std::fprintf(stderr, "single: link lane %d f=(%g %g %g) n=(%g %g %g)\n", i, (double)cfl[0].v[i], (double)cfl[1].v[i], (double)cfl[2].v[i], (double)cfa[0].v[i], (double)cfa[1].v[i], (double)cfa[2].v[i]);
#endif
            if (kRK4) {
  #pragma unroll
              for (int k = 0; k < 3; ++k) xcfl[k] = cfl[k], xcfa[k] = cfa[k];
            }
          }
          if constexpr (kDyn) {
  #pragma unroll
            for (int k = 0; k < 3; ++k) dcfl[k] = cfl[k], dcfa[k] = cfa[k];
          }
          V pAr[1][6], ar[1][6], sddr[1];
  #pragma unroll
          for (int k = 0; k < 3; ++k) pAr[0][k] = -cfl[k], pAr[0][3 + k] = -cfa[k];
          response<1>(lane, level, parent, child, tf, pAr, ar, sddr);
          sdd = sdd + sddr[0];
  #pragma unroll
          for (int k = 0; k < 3; ++k) {
            acl[k] = acl[k] + (P.floating ? ln.shfl(ar[0][k], zero_lane) : V(T(0)));
            aca[k] = aca[k] + (P.floating ? ln.shfl(ar[0][3 + k], zero_lane) : V(T(0)));
          }
        } else {
          // velocity reset at impacts: nu+ = nu - M^-1 J^T lambda on the new configuration
          V dv0[6], dsd;
          rigid_impact(lane, level, parent, child, tf, ps0, rp, mass, cw, Ic, dv0, dsd);
          sd = sd + vsel(is_joint, dsd, V(T(0)));
          V t[3];
          cross(dv0 + 3, pB, t);  // inertial-fixed linear velocity: v_W = v_C - w x p_B
  #pragma unroll
          for (int k = 0; k < 3; ++k) {
            vW[k] = vW[k] + dv0[k] - t[k];
            om[k] = om[k] + dv0[3 + k];
          }
        }
      }
    }

    ln.stamp(A, 9);  // pass 3

    if (MODE == MODE_FD) {
      // inertial-fixed base acceleration: a_lin^W = a_lin^C - wdot x p_B
      V t[3];
      cross(aca, pB, t);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        ln.gstore(A.out_a, lane * 0 + k, acl[k] - t[k], is_root, 6 + P.n);
        ln.gstore(A.out_a, lane * 0 + (3 + k), aca[k], is_root, 6 + P.n);
      }
      ln.gstore(A.out_a, jrow + 6, sdd, is_joint, 6 + P.n);
      return;
    }

    if constexpr (kDyn) {
      // ---- the outputs of system_dynamics (api/ode.py:174-225), inertial-fixed: position derivatives of
      // system_position_dynamics (:134-171: pdot_B = v_W + w x p_B, Qdot with the caller's Baumgarte gain, sdot) and the
      // accelerations above, written in the layout of the state block (row of x -> d x / dt) -------------------------------
      if (kRigid && A.out_H != nullptr) store_link_contact_wrenches();  // (the rigid models refer their wrenches to the origin of C: ra = 0)
      if (A.state_out != nullptr) {
        V dq[4], t[3];
        quat_derivative(q, om, A.fparam, dq);
        cross(aca, pB, t);
        s = sd, sd = sdd;
#pragma unroll
        for (int k = 0; k < 4; ++k) q[k] = dq[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          pB[k] = vBc[k];
          vW[k] = acl[k] - t[k];
          om[k] = aca[k];
          if (with_contacts) ps0.m[k] = ps0.md[k];
        }
        write_state(A.state_out, 0, P.n_rows);
      }
      return;
    } else if (kRigid && stage == kImpactStage) {
      // impact stage: the velocities were reset above, nothing to integrate
    } else if (!kRK4) {
    // ---- C: semi-implicit Euler (api/integrators.py:14-88) ---------------------------------
      const V dt = V(P.dt);
      sd = sd + dt * sdd;
      s = s + dt * sd;
      // base: w+ = w + dt wdot ; pdot = (v_W + dt a_W) + w+ x p_B = vBc + dt a_lin^C
      V omn[3], pd[3], t[3];
      cross(aca, pB, t);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        omn[k] = om[k] + dt * aca[k];
        pd[k] = vBc[k] + dt * acl[k];
        vW[k] = vW[k] + dt * (acl[k] - t[k]);
      }
      // Qdot = 1/2 Q_inertial(q) [K |w| (1 - |q|); w]   (math/quaternion.py:68-132)
      V qd[4];
      quat_derivative(q, omn, P.quat_K, qd);
      V qn[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) qn[k] = q[k] + dt * qd[k];
      const V nn = vsqrt(qn[0] * qn[0] + qn[1] * qn[1] + qn[2] * qn[2] + qn[3] * qn[3]);
      const V invn = vrcp(vsel(nn == V(T(0)), V(T(1)), nn));
#pragma unroll
      for (int k = 0; k < 4; ++k) q[k] = qn[k] * invn;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        pB[k] = pB[k] + dt * pd[k];
        om[k] = omn[k];
        if (with_contacts) ps0.m[k] = ps0.m[k] + dt * ps0.md[k];  // api/integrators.py:67-71
      }
    } else {
      // ---- system_dynamics at this stage (api/ode.py:134-225): position derivatives use the stage
      // velocity, the quaternion derivative the Baumgarte gain 1.0 ------------------------------------
      V dq[4], dv[3], t[3], pdt[3];
      quat_derivative(q, om, T(1), dq);
      cross(aca, pB, t);
#pragma unroll
      for (int k = 0; k < 3; ++k) dv[k] = acl[k] - t[k], pdt[k] = vBc[k];  // inertial-fixed linear acceleration; W_pdot_B = v_W + w x p_B
      V sd_st = sd;
      if (kRigid && P.rk4fast) {
        // RungeKutta4Fast: system_position_dynamics is evaluated on the initial data at every stage
        // (integrators.py:200-203): the position derivatives of stage 0 are reused
        if (stage == 0) {
          sd0 = sd;
#pragma unroll
          for (int k = 0; k < 4; ++k) dq0[k] = dq[k];
#pragma unroll
          for (int k = 0; k < 3; ++k) pd0[k] = pdt[k];
        }
        sd_st = sd0;
#pragma unroll
        for (int k = 0; k < 4; ++k) dq[k] = dq0[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) pdt[k] = pd0[k];
      }
      const T wgt = (stage == 0 || stage == 3) ? T(1) : T(2);
      ks = ks + wgt * sd_st, ksd = ksd + wgt * sdd;
#pragma unroll
      for (int k = 0; k < 4; ++k) kq[k] = kq[k] + wgt * dq[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        kp[k] = kp[k] + wgt * pdt[k];
        kv[k] = kv[k] + wgt * dv[k];
        kw[k] = kw[k] + wgt * aca[k];
        if (with_contacts) km[k] = km[k] + wgt * ps0.md[k];
      }
      if (stage < 3) {
        const V h = V(stage == 2 ? P.dt : P.dt * T(0.5));  // euler_mid, euler_mid, euler_fin
        const V sdd_st = sdd;
        s = x0s + h * sd_st;
        sd = x0sd + h * sdd_st;
#pragma unroll
        for (int k = 0; k < 4; ++k) q[k] = x0q[k] + h * dq[k];  // re-normalised at the next stage start
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          pB[k] = x0p[k] + h * pdt[k];
          vW[k] = x0v[k] + h * dv[k];
          om[k] = x0w[k] + h * aca[k];
          if (with_contacts) ps0.m[k] = x0m[k] + h * ps0.md[k];
        }
      } else {
        const V h = V(P.dt * T(1.0 / 6.0));
        s = x0s + h * ks;
        sd = x0sd + h * ksd;
        V qn[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) qn[k] = x0q[k] + h * kq[k];
        const V nn = vsqrt(qn[0] * qn[0] + qn[1] * qn[1] + qn[2] * qn[2] + qn[3] * qn[3]);
        const V invn = vrcp(vsel(nn == V(T(0)), V(T(1)), nn));  // data.replace (api/data.py:434-440)
#pragma unroll
        for (int k = 0; k < 4; ++k) q[k] = qn[k] * invn;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          pB[k] = x0p[k] + h * kp[k];
          vW[k] = x0v[k] + h * kv[k];
          om[k] = x0w[k] + h * kw[k];
          if (with_contacts) ps0.m[k] = x0m[k] + h * km[k];
        }
      }
    }
    }  // integrator stages
    // [round 4] recorded rollout (jxs_rollout_recorded): the state after EVERY step goes to the trajectory block
    // [n_steps * n_rows][N] (rows it * n_rows ...), what jax.lax.scan over step returns as its stacked outputs.  The
    // pointer travels in KArgs::out_a, which the step modes do not use otherwise.
    if (MODE == MODE_ROLLOUT && A.out_a != nullptr) write_state(A.out_a, it * P.n_rows, n_steps * P.n_rows);
    }  // fused step loop

    write_state(A.state_out, 0, P.n_rows);
    ln.stamp(A, 10);  // integrate + stores issued
  }

  // ---- pieces of run() shared with the inertia wave of a two-wave workgroup (run_inertia) --------------
  // base rotation: q <- q / |q| and its DCM (data.base_orientation, api/data.py:267-286)
  JXS_HD void base_dcm(V* q, V* R0) const {
    {
      // q / (|q| + eps where |q| = 0)  (data.base_orientation): 1 / |q| as ONE refined reciprocal square root -- this is the
      // head of the step's critical path (square root, guard, reciprocal: twelve dependent instructions before)
      const V nsq = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
      const V inv = vsel(nsq > V(T(0)), vrsqrt(vsel(nsq > V(T(0)), nsq, V(T(1)))), V(T(1)));  // (|q| = 0: q / eps = 0 = q)
#pragma unroll
      for (int k = 0; k < 4; ++k) q[k] = q[k] * inv;
    }
    const V w = q[0], x = q[1], y = q[2], z = q[3];
    const V two = V(T(2));
    R0[0] = V(T(1)) - two * (y * y + z * z);
    R0[1] = two * (x * y - w * z);
    R0[2] = two * (x * z + w * y);
    R0[3] = two * (x * y + w * z);
    R0[4] = V(T(1)) - two * (x * x + z * z);
    R0[5] = two * (y * z - w * x);
    R0[6] = two * (x * z - w * y);
    R0[7] = two * (y * z + w * x);
    R0[8] = V(T(1)) - two * (x * x + y * y);
  }
  // local parent->child transform lambda_H_pre * pre_H_suc(s) * suc_H_i; the base lane starts from (R0, 0)
  JXS_HD void local_transform(const VM& is_rev, const VM& is_pri, const VM& is_root, const V& s, const V* ax, const V* Rpre,
                              const V* ppre, const V* Rsuc, const V* psuc, const V* R0, V* R, V* r) const {
    // Rodrigues from the half angle: sin s = 2 sh ch, 1 - cos s = 2 sh^2 (the reference also
    // forms 1 - cos as 2 sin^2(theta/2)), cos s = 1 - 2 sh^2.
    V sh, chh;
    vsincos((P.any_pri ? vsel(is_rev, s, V(T(0))) : s) * T(0.5), sh, chh);  // (s is zero in the lanes without a joint)
    const V sn = T(2) * sh * chh;
    const V c1 = T(2) * sh * sh;
    const V cs = V(T(1)) - c1;
    V Rj[9];
    Rj[0] = cs + c1 * ax[0] * ax[0];
    Rj[1] = -sn * ax[2] + c1 * ax[0] * ax[1];
    Rj[2] = sn * ax[1] + c1 * ax[0] * ax[2];
    Rj[3] = sn * ax[2] + c1 * ax[1] * ax[0];
    Rj[4] = cs + c1 * ax[1] * ax[1];
    Rj[5] = -sn * ax[0] + c1 * ax[1] * ax[2];
    Rj[6] = -sn * ax[1] + c1 * ax[2] * ax[0];
    Rj[7] = sn * ax[0] + c1 * ax[2] * ax[1];
    Rj[8] = cs + c1 * ax[2] * ax[2];
    V pj[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) pj[k] = P.any_pri ? vsel(is_pri, s * ax[k], V(T(0))) : V(T(0));
    V Rl[9], pl[3];
    if (P.any_suc) {
      V tmp[9], t3[3];
      mat3mul(Rj, Rsuc, tmp);
      mat3mul(Rpre, tmp, Rl);
      mat3vec(Rj, psuc, t3);
#pragma unroll
      for (int k = 0; k < 3; ++k) t3[k] = t3[k] + pj[k];
      mat3vec(Rpre, t3, pl);
    } else {
      mat3mul(Rpre, Rj, Rl);
      if (P.any_pri) mat3vec(Rpre, pj, pl);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) pl[k] = (P.any_pri || P.any_suc) ? pl[k] + ppre[k] : ppre[k];  // (no prismatic joint: lambda_H_pre's translation)
    // The base lane starts from (R0, 0): frame C has its origin at the base position.
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = vsel(is_root, R0[k], Rl[k]);
#pragma unroll
    for (int k = 0; k < 3; ++k) r[k] = vsel(is_root, V(T(0)), pl[k]);
  }
  // forward kinematics as a tree prefix product (pointer jumping)
  JXS_HD void fk_prefix(const VI* jump, V* R, V* r) const {
#pragma unroll
    for (int k = 0; k < kMaxRounds; ++k) {
      if (k < P.n_rounds) {
        const VI src = jump[k];
        const VM ok = src >= 0;
        V Ra[9], ra[3];
#pragma unroll
        for (int e = 0; e < 9; ++e) Ra[e] = ln.shfl(R[e], src);
#pragma unroll
        for (int e = 0; e < 3; ++e) ra[e] = ln.shfl(r[e], src);
        ln.fence();
        V Rn[9], rn[3];
        mat3mul(Ra, R, Rn);
        mat3vec(Ra, r, rn);
        if (P.jump_pad) {  // sources beyond the base are a padding lane with the identity transform: no select
#pragma unroll
          for (int e = 0; e < 9; ++e) R[e] = Rn[e];
#pragma unroll
          for (int e = 0; e < 3; ++e) r[e] = rn[e] + ra[e];
        } else {
#pragma unroll
          for (int e = 0; e < 9; ++e) R[e] = vsel(ok, Rn[e], R[e]);
#pragma unroll
          for (int e = 0; e < 3; ++e) r[e] = vsel(ok, rn[e] + ra[e], r[e]);
        }
      }
    }
  }
  // CoM c = r + R c_L and rotational inertia I_c = R I_L R^T of the link in C
  JXS_HD void link_inertia_C(const V* R, const V* r, const V* cL, const V* IL, V* cw, V* Ic) const {
    mat3vec(R, cL, cw);
#pragma unroll
    for (int k = 0; k < 3; ++k) cw[k] = cw[k] + r[k];
    // T = R * I_L (3x3), then Ic = T * R^T (symmetric)
    const V I9[9] = {IL[0], IL[1], IL[2], IL[1], IL[3], IL[4], IL[2], IL[4], IL[5]};
    V Tm[9];
    mat3mul(R, I9, Tm);
    Ic[0] = Tm[0] * R[0] + Tm[1] * R[1] + Tm[2] * R[2];
    Ic[1] = Tm[0] * R[3] + Tm[1] * R[4] + Tm[2] * R[5];
    Ic[2] = Tm[0] * R[6] + Tm[1] * R[7] + Tm[2] * R[8];
    Ic[3] = Tm[3] * R[3] + Tm[4] * R[4] + Tm[5] * R[5];
    Ic[4] = Tm[3] * R[6] + Tm[4] * R[7] + Tm[5] * R[8];
    Ic[5] = Tm[6] * R[6] + Tm[7] * R[7] + Tm[8] * R[8];
  }
  // M = [[m I, m S(c)^T],[m S(c), I_c + m S(c) S(c)^T]]  (math/inertia.py:14-41), upper triangle
  static JXS_HD void assemble_MA(const V& mass, const V* cw, const V* Ic, V* MA) {
    const V zero = V(T(0));
    const V mcx = mass * cw[0], mcy = mass * cw[1], mcz = mass * cw[2];
    // top-left: m I
    MA[sidx(0, 0)] = mass; MA[sidx(0, 1)] = zero; MA[sidx(0, 2)] = zero;
    MA[sidx(1, 1)] = mass; MA[sidx(1, 2)] = zero;
    MA[sidx(2, 2)] = mass;
    // top-right: m S(c)^T = -m S(c) = [[0, mcz, -mcy],[-mcz, 0, mcx],[mcy, -mcx, 0]]
    MA[sidx(0, 3)] = zero; MA[sidx(0, 4)] = mcz; MA[sidx(0, 5)] = -mcy;
    MA[sidx(1, 3)] = -mcz; MA[sidx(1, 4)] = zero; MA[sidx(1, 5)] = mcx;
    MA[sidx(2, 3)] = mcy; MA[sidx(2, 4)] = -mcx; MA[sidx(2, 5)] = zero;
    // bottom-right: I_c + m (|c|^2 I - c c^T)
    const V cc = cw[0] * cw[0] + cw[1] * cw[1] + cw[2] * cw[2];
    MA[sidx(3, 3)] = Ic[0] + mass * (cc - cw[0] * cw[0]);
    MA[sidx(3, 4)] = Ic[1] - mcx * cw[1];
    MA[sidx(3, 5)] = Ic[2] - mcx * cw[2];
    MA[sidx(4, 4)] = Ic[3] + mass * (cc - cw[1] * cw[1]);
    MA[sidx(4, 5)] = Ic[4] - mcy * cw[2];
    MA[sidx(5, 5)] = Ic[5] + mass * (cc - cw[2] * cw[2]);
  }

  // Qdot = 1/2 Q_inertial(q) [K |w| (1 - |q|); w]   (math/quaternion.py:68-132), q normalised
  static JXS_HD void quat_derivative(const V* q, const V* w, const T K, V* qd) {
    const V nw = vsqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    const V nq = vsqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const V h0 = K * nw * (V(T(1)) - nq);
    const V half = V(T(0.5));
    qd[0] = half * (q[0] * h0 - q[1] * w[0] - q[2] * w[1] - q[3] * w[2]);
    qd[1] = half * (q[1] * h0 + q[0] * w[0] + q[3] * w[1] - q[2] * w[2]);
    qd[2] = half * (q[2] * h0 - q[3] * w[0] + q[0] * w[1] + q[1] * w[2]);
    qd[3] = half * (q[3] * h0 + q[2] * w[0] - q[1] * w[1] + q[0] * w[2]);
  }

  // ==========================================================================================
  // inclusive prefix sum of a 6-vector over the ancestors of every lane (pointer jumping)
  JXS_HD void prefix6(const VI* jump, V* xl, V* xa) const {
#pragma unroll
    for (int k = 0; k < kMaxRounds; ++k) {
      if (k < P.n_rounds) {
        const VI src = jump[k];
        const VM ok = src >= 0;
        V tl[3], ta[3];
#pragma unroll
        for (int e = 0; e < 3; ++e) {
          tl[e] = ln.shfl(xl[e], src);
          ta[e] = ln.shfl(xa[e], src);
        }
        ln.fence();
        if (P.jump_pad) {  // (sources beyond the base are a padding lane holding zeros)
#pragma unroll
          for (int e = 0; e < 3; ++e) xl[e] = xl[e] + tl[e], xa[e] = xa[e] + ta[e];
        } else {
#pragma unroll
          for (int e = 0; e < 3; ++e) {
            xl[e] = xl[e] + vsel(ok, tl[e], V(T(0)));
            xa[e] = xa[e] + vsel(ok, ta[e], V(T(0)));
          }
        }
      }
    }
  }

  // Pivots K..5 of the Gauss-Jordan elimination of a 6x6 system whose row r (x[0..5]) and right-hand side entry (x[6])
  // sit in lane r of a 16-lane row: afterwards x[6] / x[r] is the solution entry of lane r (aba_rows, base solve).
  // Columns <= K of the other rows are not updated: nothing reads them again.  The chain from pivot to pivot is
  // broadcast -> reciprocal -> multiplier -> fused update (four dependent instructions): the multiplier takes the plain
  // reciprocal (1 ulp: a backward error of one ulp in column K) and the own-row mask as a factor prepared beside the chain.
  template <int K>
  JXS_HD void gauss_jordan_rows(const VI& row, V* x) const {
    if constexpr (K < 6) {
      const V xm = vsel(row == K, V(T(0)), -x[K]);
      const V nf = xm * vrcp(ln.row_bcast(x[K], K));
      ln.template fmac_row_bcast<K, 6 - K>(x + (K + 1), nf);
      gauss_jordan_rows<K + 1>(row, x);
    }
  }

  // a = -MA^-1 pA for the symmetric positive-definite 6x6 MA (LDL^T, no pivoting)
  JXS_HD void solve6(const V* MA, const V* pA, V* a) const {
    V Lm[6][6], Dd[6], Di[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      V dj = MA[sidx(j, j)];
#pragma unroll
      for (int k = 0; k < j; ++k) dj = dj - Lm[j][k] * Lm[j][k] * Dd[k];
      Dd[j] = dj;
      Di[j] = vrcp_acc(dj);
#pragma unroll
      for (int i = j + 1; i < 6; ++i) {
        V lij = MA[sidx(i, j)];
#pragma unroll
        for (int k = 0; k < j; ++k) lij = lij - Lm[i][k] * Lm[j][k] * Dd[k];
        Lm[i][j] = lij * Di[j];
      }
    }
    V y[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      V acc = -pA[i];
#pragma unroll
      for (int k = 0; k < i; ++k) acc = acc - Lm[i][k] * y[k];
      y[i] = acc;
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
      V acc = y[i] * Di[i];
#pragma unroll
      for (int k = i + 1; k < 6; ++k) acc = acc - Lm[k][i] * a[k];
      a[i] = acc;
    }
  }

  // ==========================================================================================
  // Soft contacts: point kinematics (rbda/collidable_points.py:9-65), penetration
  // (rbda/contacts/common.py:25-63), Hunt-Crossley + stick/slip state
  // (rbda/contacts/soft.py:195-388), per-link wrench sum (api/contact.py:557-603) and the
  // Euler update of the tangential deformation (api/integrators.py:67-71).
  // ==========================================================================================
  // Row-distributed ABA passes.  Pass 2 in the link-per-lane layout spends
  // ~180 VALU per tree level with one lane in eight doing useful work (measured: 42 % of the
  // step).  Here lane = 8*slot + row: the 8 lanes of a slot hold the rows of the articulated
  // inertia of the ONE link that slot owns at the current level; link lanes publish
  // (M, S, c, pA, tau) once through LDS, row lanes read their row per level.  A first child
  // inherits its parent's slot, so along a chain "propagate to the parent" is a register add in
  // the same lane; only extra children cross slots (one 7-value shuffle batch).  U = MA S is a
  // column all-reduce over the 8 lanes (3 DPP adds per entry, MA symmetric).
  struct RowTabs {
    VI rec[kRowLevels], ppull[kRowLevels], pull[kRowLevels][kRowExtra], fcbits;
  };
  JXS_HD void load_row_tabs(RowTabs& rt) const {
#pragma unroll
    for (int Lv = 0; Lv < kRowLevels; ++Lv) {
      rt.rec[Lv] = ln.rconsti(A.rti, RT_REC + Lv);
      rt.ppull[Lv] = ln.rconsti(A.rti, RT_PPULL + Lv);
#pragma unroll
      for (int k = 0; k < kRowExtra; ++k) rt.pull[Lv][k] = ln.rconsti(A.rti, RT_PULL + Lv * kRowExtra + k);
    }
    rt.fcbits = ln.rconsti(A.rti, RT_FC);
  }

  struct RowLevel {
    V Mrow[6], S[6], c[6], pr, S_r, c_r, tau, dp[3];
  };
  // [round 6] address of the 8-word row group a row lane reads of record `rec`: +8 row in a link's record; a lane without
  // a link reads zeros at +48 of lds_zero_rec (jxs_params.h: the zero area is 16 words, not a whole record)
  JXS_HD VI row_group(const VI& rec, const VI& row6) const { return rec + vsel(rec != lds_zero_rec(P.nL), row6 * 8, row6 * 0 + RL_S); }
  // record of this lane's link; lanes without a link (lane >= nL) publish nothing and read their result from the zero area
  JXS_HD VI own_record(const VI& lane) const { return vsel(lane < P.nL, lane * kRowRec + ((lane + 1) >> 1) * 4, lane * 0 + lds_zero_rec(P.nL)); }  // (lds_rec_off)
  // the zero area [Z, Z + kRowZero): kRowZero / 4 lanes write four words each
  JXS_HD void write_zero_area(const VI& lane, int area) const {
    const V z4[4] = {V(T(0)), V(T(0)), V(T(0)), V(T(0))};
    static_assert(kRowZero % 4 == 0 && kRowZero / 4 <= 8, "zero area: whole 16-byte groups, written by the first lanes of the group (the row layout needs 8 lanes or more)");
    ln.template lds_writev_if<4>(lane * 4 + (lds_zero_at(P.nL) + area), z4, lane < kRowZero / 4);
  }
  JXS_HD void load_row_level(const VI& rec, const VI& row6, const VI& lane, RowLevel& o) const {
    // `rec`: the link's record for its six row lanes; the all-zero record for lanes without a link at this
    // level (empty slot, idle lanes 6 and 7 of a slot) -- resolved by the packer, no address selects here
    (void)lane;
    const VI base = rec;
    V rw[8], sc[12], td[4];
    ln.template lds_readv<8>(row_group(rec, row6), rw);
    ln.template lds_readv<12>(base + RL_S, sc);
    ln.template lds_readv<4>(base + RL_TAU, td);
    o.c_r = ln.lds_read(base + row6 + RL_C);
#pragma unroll
    for (int j = 0; j < 6; ++j) o.Mrow[j] = rw[j], o.S[j] = sc[j], o.c[j] = sc[6 + j];
    o.pr = rw[RL_ROW_PA], o.S_r = rw[RL_ROW_S];
    o.tau = td[0];
#pragma unroll
    for (int k = 0; k < 3; ++k) o.dp[k] = td[1 + k];
  }

  JXS_HD void aba_rows(const VI& lane, const RowTabs& rt, const V* MA, const V* pA, const V* S6, const V* c6,
                       const V& tau, const bool anch, const V* dpl, V& sdd, V* a0) const {
    const V zero = V(T(0));
    const VI rec_me = own_record(lane);
    // ---- link lanes publish their record (128-bit writes; the lanes behind the last link publish nothing) ------------
    ln.lds_masked(lane < P.nL, [&]() {
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const V rw[8] = {MA[sidx(i, 0)], MA[sidx(i, 1)], MA[sidx(i, 2)], MA[sidx(i, 3)], MA[sidx(i, 4)], MA[sidx(i, 5)], pA[i], S6[i]};
        ln.template lds_writev<8>(rec_me + (RL_ROW + 8 * i), rw);
      }
      const V sc[12] = {S6[0], S6[1], S6[2], S6[3], S6[4], S6[5], c6[0], c6[1], c6[2], c6[3], c6[4], c6[5]};
      ln.template lds_writev<12>(rec_me + RL_S, sc);
      const V td[4] = {tau, anch ? dpl[0] : zero, anch ? dpl[1] : zero, anch ? dpl[2] : zero};
      ln.template lds_writev<4>(rec_me + RL_TAU, td);
    });
    write_zero_area(lane, 0);
    const int XB = lds_base_rows(G);  // base rows: [XB + 8 r + j], j < 6 inertia row, j = 6 bias force (the row region of record 0)

    const VI row = lane & 7;
    const VM rowok = row < 6;
    const VI row6 = vsel(rowok, row, lane * 0);
    const int max_depth = P.max_depth;
    // anchored chains: where a non-first child joins its parent, the child's slot re-refers its rows by
    // d = anchor(child chain) - anchor(parent chain) before they are pulled (xlate_inertia in row form):
    //   every row:   row[3:6] -= row[0:3] x d
    //   angular row j (lane row 3 + j):  row += d_{j+1} LIN_{j+2} - d_{j+2} LIN_{j+1}   (LIN_k = linear row k)
    // The two linear rows an angular lane needs sit in the same slot: two shuffle sources per lane.
    const VM is_ang = (row >= 3) && rowok, is_lin = row < 3;
    const VI jj = vsel(is_ang, row - 3, row);              // index within the linear / angular triple
    const VI j1 = vsel(jj == 2, lane * 0, jj + 1);          // (j + 1) % 3
    const VI j2 = vsel(jj == 0, lane * 0 + 2, jj - 1);      // (j + 2) % 3
    const VI slot0 = lane - row;                            // first lane of this slot
    const VI xsrc1 = slot0 + j1, xsrc2 = slot0 + j2;        // linear rows (j+1)%3, (j+2)%3 of this slot
    auto pick3 = [&](const V* d3, const VI& idx) { return vsel(idx == 0, d3[0], vsel(idx == 1, d3[1], d3[2])); };

    // ---- pass 2, leaves to base (rbda/aba.py:184-224) -------------------------------------
    // Wave-uniform flags are pinned in SGPRs up front (the compiler otherwise re-loads them from
    // the kernel argument inside every level: one ~200-cycle scalar-load stall per level), and
    // the LDS reads of level L-1 are issued before level L is computed (software pipelining).
    const unsigned cross_levels = ln.pin(P.row_cross_levels);
    const unsigned ppull_levels = ln.pin(P.row_ppull_levels);
    const unsigned pull_counts = ln.pin(P.row_pull_counts);
    const int floating = ln.pin(P.floating);
    V Ur[kRowLevels], Sr[kRowLevels], cr[kRowLevels], invd[kRowLevels], uu[kRowLevels];
    V accM[6], accp = zero, MA0[6], p0 = zero;
    V dkeep[kRowLevels][3];  // anchored chains: d of the links that leave their chain at a level (pass 2 -> pass 3)
#pragma unroll
    for (int j = 0; j < 6; ++j) accM[j] = zero, MA0[j] = zero;
    RowLevel cur, nxt;
    load_row_level(rt.rec[kRowLevels - 1], row6, lane, cur);
#pragma unroll
    for (int Lv = kRowLevels - 1; Lv >= 0; --Lv) {
      Ur[Lv] = zero, Sr[Lv] = zero, cr[Lv] = zero, invd[Lv] = zero, uu[Lv] = zero;
      if (Lv >= 1) load_row_level(rt.rec[Lv - 1], row6, lane, nxt);  // prefetch the next level
      if (Lv <= max_depth && (Lv >= 1 || floating)) {
        const VM has = rt.rec[Lv] != lds_zero_rec(P.nL);
        V MArow[6];  // (lanes without a link, and the idle lanes 6, 7 of a slot, read the zero area)
        if (Lv == max_depth) {  // (the deepest level: nothing has been handed up yet -- IEEE arithmetic keeps x + 0)
#pragma unroll
          for (int j = 0; j < 6; ++j) MArow[j] = cur.Mrow[j];
        } else if (!L::add6_packed(cur.Mrow, accM, MArow)) {
#pragma unroll
          for (int j = 0; j < 6; ++j) MArow[j] = cur.Mrow[j] + accM[j];
        }
        const V pr = (Lv == max_depth) ? cur.pr : cur.pr + accp;
        const V S_r = cur.S_r;
        const V c_r = cur.c_r;
        if (Lv == 0) {
#pragma unroll
          for (int j = 0; j < 6; ++j) MA0[j] = MArow[j];
          p0 = pr;
        } else {
          // U_r = (MA S)_r in the lane of row r; d = S.U and S.pA are 8-lane reductions of per-lane products; the full U
          // is never gathered: the rank-one update Ma = MA - U U^T / d takes U_j straight out of lane j of the slot
          // (ln.rank1_rows).  [round 3; before: U in every lane by six reductions of the scaled rows, 21 + 3 + 8 instructions]
          V U_r;
          if (!L::dot6_packed(MArow, cur.S, &U_r)) {
            U_r = MArow[0] * cur.S[0];
#pragma unroll
            for (int j = 1; j < 6; ++j) U_r = U_r + MArow[j] * cur.S[j];
          }
          V red[2] = {S_r * U_r, S_r * pr};
          ln.allreduce8x2(red);
          const V d = red[0];
          const V u = cur.tau - red[1];
          // -1 / d, zero in the lanes without a link (their d = 0: whatever the reciprocal returns there is dropped by the
          // select, not multiplied away).  The sign lives in the select: no negation on the chain to the rank-one update.
          const V ninv = vsel(has, -vrcp_acc(d), zero);
          const V nUd = U_r * ninv;
          V Ma[6], pa = pr - nUd * u;
#pragma unroll
          for (int j = 0; j < 6; ++j) Ma[j] = MArow[j];
          ln.rank1_rows(Ma, U_r, nUd);
          {
            V t;
            if (L::dot6_packed(Ma, cur.c, &t)) {
              pa = pa + t;
            } else {
#pragma unroll
              for (int j = 0; j < 6; ++j) pa = pa + Ma[j] * cur.c[j];
            }
          }
          Ur[Lv] = U_r, Sr[Lv] = S_r, cr[Lv] = c_r, invd[Lv] = ninv, uu[Lv] = u;  // (invd: -1 / d)
          // propagate: first children stay in their lanes, extra children are pulled by the
          // parent's lanes.  A fixed base receives nothing (rbda/aba.py:217-222).
          if (Lv >= 2 || floating) {
            if (anch && ((cross_levels >> Lv) & 1u)) {
              // re-refer the rows of the links that leave their chain here (d = 0 for first children)
              V d3[3];  // (arrived with tau; zero where the slot holds no link)
#pragma unroll
              for (int k = 0; k < 3; ++k) d3[k] = cur.dp[k], dkeep[Lv][k] = d3[k];
              {  // row[3:6] -= row[0:3] x d
                const V t0 = Ma[1] * d3[2] - Ma[2] * d3[1], t1 = Ma[2] * d3[0] - Ma[0] * d3[2], t2 = Ma[0] * d3[1] - Ma[1] * d3[0];
                Ma[3] = Ma[3] - t0, Ma[4] = Ma[4] - t1, Ma[5] = Ma[5] - t2;
              }
              const V c1 = vsel(is_ang, pick3(d3, j1), zero);    //  d_{j+1}: multiplies LIN_{j+2}
              const V c2 = vsel(is_ang, -pick3(d3, j2), zero);   // -d_{j+2}: multiplies LIN_{j+1}
              V x7[7] = {Ma[0], Ma[1], Ma[2], Ma[3], Ma[4], Ma[5], pa};
              ln.ang_from_lin7(x7, c1, c2, xsrc1, xsrc2);  // x += c1 * x@(linear row j + 2) + c2 * x@(linear row j + 1)
#pragma unroll
              for (int j = 0; j < 6; ++j) Ma[j] = x7[j];
              pa = x7[6];
            }
            // what a first child hands to its parent in the same lanes: x 1, else x 0 (values are finite: packed multiplies
            // instead of seven selects)
            const V fcf = vsel(((rt.fcbits >> Lv) & 1) != 0, V(T(1)), zero);
            if (!L::scale6_packed(Ma, fcf, accM)) {
#pragma unroll
              for (int j = 0; j < 6; ++j) accM[j] = Ma[j] * fcf;
            }
            accp = pa * fcf;
            if ((cross_levels >> Lv) & 1u) {
              const int npull = (int)((pull_counts >> (4 * Lv)) & 15u);  // wave-uniform
#pragma unroll
              for (int k = 0; k < kRowExtra; ++k) {
                if (k >= npull) break;
                const VI src = rt.pull[Lv][k];  // (nothing to pull: an idle row lane, whose values are zero)
                if (L::kHasRowShift && ((P.row_pull_dpp >> (Lv * kRowExtra + k)) & 1u)) {
                  // the child sits in the next slot of the same 16-lane row: a DPP row shift (jxs_lanes_device.h)
                  V acc7[7] = {accM[0], accM[1], accM[2], accM[3], accM[4], accM[5], accp};
                  const V x7[7] = {Ma[0], Ma[1], Ma[2], Ma[3], Ma[4], Ma[5], pa};
                  ln.fmac7_from_next_slot(acc7, x7, vsel(src != 6, V(T(1)), zero));
#pragma unroll
                  for (int j = 0; j < 6; ++j) accM[j] = acc7[j];
                  accp = acc7[6];
                  continue;
                }
                V g[7];
#pragma unroll
                for (int j = 0; j < 6; ++j) g[j] = ln.shfl(Ma[j], src);
                g[6] = ln.shfl(pa, src);
                ln.fence();
#pragma unroll
                for (int j = 0; j < 6; ++j) accM[j] = accM[j] + g[j];
                accp = accp + g[6];
              }
            }
          }
        }
      }
      cur = nxt;
      ln.stamp(A, 24 + Lv);  // (profiling build: one stamp per tree level)
    }

    ln.stamp(A, 7);  // pass 2 (row-distributed)
    // ---- base acceleration (rbda/aba.py:240-243) --------------------------------------------
    V a0_own = zero;           // entry `row` of the base acceleration in the six row lanes of slot 0, zero elsewhere
    bool a0_in_lanes = false;
    if (floating && G >= 16 && !kNoRowBaseSolve) {
      // [round 3] Gauss-Jordan across the six row lanes of slot 0, which hold the rows of MA_0 and of pA_0 already:
      // per pivot one reciprocal, one multiplier and 6 - k fused "own -= f * (row k)" whose row-k operand is a DPP
      // row broadcast -- 60 instructions where every lane used to factor the same 6x6 (LDL^T, ~170 instructions after
      // twelve LDS reads).  No pivoting: MA_0 is symmetric positive definite.  Lanes without a base row (rows 6, 7,
      // other slots of the 16-lane row) hold zeros and keep them; the other 16-lane rows compute values nobody reads.
      V x[7];  // x[0..5]: this lane's row of MA_0, x[6]: its entry of -pA_0
#pragma unroll
      for (int j = 0; j < 6; ++j) x[j] = MA0[j];
      x[6] = -p0;
      gauss_jordan_rows<0>(row, x);
      // (own pivot: the entries that settle last are selected last)
      const V piv = vsel(row == 5, x[5], vsel(row == 4, x[4], vsel(row == 3, x[3], vsel(row == 2, x[2], vsel(row == 1, x[1], x[0])))));
      a0_own = vsel(lane < 6, x[6] * vrcp_acc(piv), zero);  // pass 3 starts from here, without the LDS round trip below
      a0_in_lanes = true;
      ln.lds_write(lane + XB, a0_own, lane < 6);
      V rw[8];
      ln.template lds_readv<8>(lane * 0 + XB, rw);
#pragma unroll
      for (int k = 0; k < 6; ++k) a0[k] = rw[k];
    } else if (floating) {
      // the six row lanes of slot 0 publish the base rows; every lane then solves the same 6x6
      {
        const V rw[8] = {MA0[0], MA0[1], MA0[2], MA0[3], MA0[4], MA0[5], p0, zero};
        ln.template lds_writev_if<8>(row6 * 8 + XB, rw, lane < 6);
      }
      V MAs[21], pAs[6];
      const VI z = lane * 0;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        V rw[8];
        ln.template lds_readv<8>(z + (XB + 8 * i), rw);
#pragma unroll
        for (int j = i; j < 6; ++j) MAs[sidx(i, j)] = rw[j];
        pAs[i] = rw[6];
      }
      solve6(MAs, pAs, a0);
    } else {
#pragma unroll
      for (int k = 0; k < 6; ++k) a0[k] = zero;
      a0[2] = V(-P.g);  // a0 = -B_X_W W_g expressed in C
    }

    ln.stamp(A, 8);  // base solve
    // ---- pass 3, base to leaves (rbda/aba.py:251-267) -----------------------------------------
    // (only slot 0 starts from the base: every other slot begins with a link that pulls its parent's acceleration)
    V acar = a0_in_lanes ? a0_own
                         : vsel(row == 0, a0[0], vsel(row == 1, a0[1], vsel(row == 2, a0[2],
                           vsel(row == 3, a0[3], vsel(row == 4, a0[4], vsel(row == 5, a0[5], zero))))));
#pragma unroll
    for (int Lv = 1; Lv < kRowLevels; ++Lv) {
      if (Lv <= max_depth) {
        V apar = acar;
        if ((ppull_levels >> Lv) & 1u) {
          const VM pulled = rt.ppull[Lv] >= 0;
          const V q = ln.shfl(acar, rt.ppull[Lv]);
          if (anch) {
            // the parent sits in another chain: its acceleration moves to this chain's anchor,
            // a_lin += alpha x d, i.e. linear row k += alpha_{k+1} d_{k+2} - alpha_{k+2} d_{k+1}
            const VI pslot = vsel(pulled, rt.ppull[Lv] - row, lane * 0);
            const V al1 = ln.shfl(acar, pslot + 3 + j1), al2 = ln.shfl(acar, pslot + 3 + j2);
            ln.fence();
            const V* d3 = dkeep[Lv];  // cross levels and parent-pull levels are the same levels
            // (the lane masks go into the coefficients, beside the chain: shuffle -> multiply -> fused multiply-add -> add)
            const VM shm = pulled && is_lin;
            const V k2 = vsel(shm, pick3(d3, j2), zero), k1 = vsel(shm, pick3(d3, j1), zero);
            apar = vsel(pulled, q, acar) + (al1 * k2 - al2 * k1);
          } else {
            ln.fence();
            apar = vsel(pulled, q, acar);
          }
        }
        const V ai = apar + cr[Lv];
        const V tot = ln.allreduce8(Ur[Lv] * ai);
        const V sd = (tot - uu[Lv]) * invd[Lv];  // (invd = -1 / d)
        // (no select: lanes without a link at this level carry c_r = S_r = 0 and 1 / d = 0, and their apar is acar)
        acar = ai + Sr[Lv] * sd;
        // every row lane of the slot holds the same sd: all of them store it to the link's record (same address, same
        // value; lanes without a link store their zero -- 1 / d = 0 -- into the zero area) -- no exec-mask bookkeeping.
        // [round 6] RL_SDD is the record's tau word: pass 2 has read it, pass 3 reads nothing from the records
        ln.lds_write(rt.rec[Lv] + RL_SDD, sd);
      }
    }
    sdd = ln.lds_read(rec_me + RL_SDD);
  }

  // ==========================================================================================
  // Two-wave workgroups.  At the benchmark size half of the chip's SIMDs hold no wave and
  // the step time is the length of ONE wave's instruction stream.  The articulated-INERTIA recursion of ABA pass 2
  // (MA -> U = MA S -> d = S.U -> Ma = MA - U U^T / d, rbda/aba.py:184-224) depends on the joint positions only;
  // the BIAS recursion (pA -> u = tau - S.pA -> pa = pA + Ma c + U u / d) on velocities and contact forces.  So a
  // second wave on another SIMD of the CU -- the inertia wave -- runs forward kinematics and the inertia recursion
  // while the main wave runs velocities, contacts and bias forces; per level it hands over row r of Ma, U_r and
  // 1 / d through the LDS (XL), and the main wave's sweep carries one reduction and one propagated value per
  // level instead of seven and seven.  The data flow is one way (inertia -> main), so the host emulation runs the
  // two roles one after the other on the same LDS image.
  // ==========================================================================================
  JXS_HD void run_inertia() const {
    const VI lane = ln.lane();
    ln.stamp(A, 0);
    ln.stamp_hwid(A, 11);
    const VI jtype = ln.lconsti(A.lti, LI_JTYPE);
    const VI level = ln.lconsti(A.lti, LI_LEVEL);
    const VI jrow = ln.lconsti(A.lti, LI_JROW);
    VI jump[kMaxRounds];
#pragma unroll
    for (int k = 0; k < kMaxRounds; ++k) jump[k] = ln.lconsti(A.lti, LI_JUMP + k);
    V q[4];
    const bool stage_base = G >= 16;  // (this wave always has its LDS area)
    V base_stage = V(T(0));
    if (stage_base) {
      base_stage = ln.gload(A.state_in, vsel(lane < 4, lane, lane * 0) + P.row_quat, P.n_rows);  // one load: lane k < 4 = row k
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) q[k] = ln.gload_u(A.state_in, P.row_quat + k, P.n_rows);
    }
    V ax[3], Rpre[9], ppre[3], cL[3], IL[6];
#pragma unroll
    for (int k = 0; k < 3; ++k) ax[k] = ln.lconstf(A.ltf, LF_AXIS + k);
#pragma unroll
    for (int k = 0; k < 9; ++k) Rpre[k] = ln.lconstf(A.ltf, LF_RPRE + k);
#pragma unroll
    for (int k = 0; k < 3; ++k) ppre[k] = ln.lconstf(A.ltf, LF_PPRE + k);
    const V mass = ln.lconstf(A.ltf, LF_MASS);
#pragma unroll
    for (int k = 0; k < 3; ++k) cL[k] = ln.lconstf(A.ltf, LF_COM + k);
#pragma unroll
    for (int k = 0; k < 6; ++k) IL[k] = ln.lconstf(A.ltf, LF_ICOM + k);
    V Rsuc[9], psuc[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) Rsuc[k] = ln.lconstf(A.ltf, LF_RSUC + k);
#pragma unroll
    for (int k = 0; k < 3; ++k) psuc[k] = ln.lconstf(A.ltf, LF_PSUC + k);
    RowTabs rt;
    load_row_tabs(rt);
    const VI jrow_c = vsel(jrow >= 0, jrow, lane * 0);
    V s = ln.gload(A.state_in, jrow_c + P.row_s, P.n_rows);
    ln.fence();
    if (stage_base) {
      ln.lds_write(lane, base_stage, lane < 4);
      ln.lds_sync();
      ln.template lds_readv<4>(lane * 0, q);
      ln.lds_sync();
    }
    const VM is_joint = jtype != 0, is_rev = jtype == 1, is_pri = jtype == 2, is_root = level == 0;
    s = vsel(is_joint, s, V(T(0)));
    ln.stamp(A, 1);
    V R[9], r[3], R0[9];
    base_dcm(q, R0);
    local_transform(is_rev, is_pri, is_root, s, ax, Rpre, ppre, Rsuc, psuc, R0, R, r);
    ln.stamp(A, 2);
    fk_prefix(jump, R, r);
    ln.stamp(A, 3);
    const bool anch = P.anchored != 0;
    V ra[3] = {V(T(0)), V(T(0)), V(T(0))}, dpl[3] = {V(T(0)), V(T(0)), V(T(0))};
    if (anch) {
      const VI anchor = ln.lconsti(A.lti, LI_ANCHOR), panchor = ln.lconsti(A.lti, LI_PANCHOR);
      V rq[3];
#pragma unroll
      for (int e = 0; e < 3; ++e) ra[e] = ln.shfl(r[e], anchor), rq[e] = ln.shfl(r[e], panchor);
#pragma unroll
      for (int e = 0; e < 3; ++e) dpl[e] = ra[e] - rq[e];
    }
    {  // hand the kinematics to the main wave
      const V rec[kDuoFkRec] = {R[0], R[1], R[2], R[3], R[4], R[5], R[6], R[7], R[8], r[0], r[1], r[2], ra[0], ra[1], ra[2],
                                dpl[0], dpl[1], dpl[2], V(T(0)), V(T(0))};
      ln.template lds_writev<kDuoFkRec>(lane * kDuoFkRec + duo_fk_off(G), rec);
      ln.flag_post(kDuoFlagFk);
    }
    ln.stamp(A, 4);
    // motion subspace in C (kin_dyn_parameters.py:239-261), referred to the lane's anchor
    V Sl[3], Sa[3], Ra_[3];
    mat3vec(R, ax, Ra_);
    {
      V rxa[3];
      cross(r, Ra_, rxa);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        if (P.any_pri) {
          Sa[k] = vsel(is_rev, Ra_[k], V(T(0)));
          Sl[k] = vsel(is_rev, rxa[k], vsel(is_pri, Ra_[k], V(T(0))));
        } else {  // every joint is revolute, and the lanes without a joint carry a zero axis in the table: nothing to mask
          Sa[k] = Ra_[k];
          Sl[k] = rxa[k];
        }
      }
    }
    V cw[3], Ic[6];
    link_inertia_C(R, r, cL, IL, cw, Ic);
    if (anch) {
      V t[3];
      cross(ra, Sa, t);
#pragma unroll
      for (int k = 0; k < 3; ++k) Sl[k] = Sl[k] - t[k], cw[k] = cw[k] - ra[k];
    }
    V MA[21];
    assemble_MA(mass, cw, Ic, MA);
    const V S6[6] = {Sl[0], Sl[1], Sl[2], Sa[0], Sa[1], Sa[2]};
    ln.stamp(A, 6);
    aba_rows_inertia(lane, rt, MA, S6, anch, dpl);
    ln.stamp(A, 10);
  }

  // lane constants of the row layout shared by the two roles
  struct RowLaneIdx {
    VI row, row6, j1, j2, xsrc1, xsrc2;
    VM rowok, is_ang, is_lin;
  };
  JXS_HD void row_lane_idx(const VI& lane, RowLaneIdx& o) const {
    o.row = lane & 7;
    o.rowok = o.row < 6;
    o.row6 = vsel(o.rowok, o.row, lane * 0);
    o.is_ang = (o.row >= 3) && o.rowok, o.is_lin = o.row < 3;
    const VI jj = vsel(o.is_ang, o.row - 3, o.row);      // index within the linear / angular triple
    o.j1 = vsel(jj == 2, lane * 0, jj + 1);              // (j + 1) % 3
    o.j2 = vsel(jj == 0, lane * 0 + 2, jj - 1);          // (j + 2) % 3
    const VI slot0 = lane - o.row;                       // first lane of this slot
    o.xsrc1 = slot0 + o.j1, o.xsrc2 = slot0 + o.j2;      // linear rows (j+1)%3, (j+2)%3 of this slot
  }
  static JXS_HD V pick3(const V* d3, const VI& idx) { return vsel(idx == 0, d3[0], vsel(idx == 1, d3[1], d3[2])); }

  // The inertia wave: aba_rows() without everything that depends on velocities or forces.
  JXS_HD void aba_rows_inertia(const VI& lane, const RowTabs& rt, const V* MA, const V* S6, const bool anch, const V* dpl) const {
    const V zero = V(T(0));
    const VI rec_me = own_record(lane);
    ln.lds_masked(lane < P.nL, [&]() {
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const V rw[8] = {MA[sidx(i, 0)], MA[sidx(i, 1)], MA[sidx(i, 2)], MA[sidx(i, 3)], MA[sidx(i, 4)], MA[sidx(i, 5)], zero, S6[i]};
        ln.template lds_writev<8>(rec_me + (RL_ROW + 8 * i), rw);
      }
      const V sc[6] = {S6[0], S6[1], S6[2], S6[3], S6[4], S6[5]};
      ln.template lds_writev<6>(rec_me + RL_S, sc);
      const V td[4] = {zero, anch ? dpl[0] : zero, anch ? dpl[1] : zero, anch ? dpl[2] : zero};
      ln.template lds_writev<4>(rec_me + RL_TAU, td);
    });
    write_zero_area(lane, 0);
    const int XB = lds_base_rows(G);
    const int XL = duo_xl_off(G);
    RowLaneIdx ix;
    row_lane_idx(lane, ix);
    const int max_depth = P.max_depth;
    const unsigned cross_levels = ln.pin(P.row_cross_levels);
    const unsigned pull_counts = ln.pin(P.row_pull_counts);
    const int floating = ln.pin(P.floating);
    struct Lvl {
      V Mrow[6], S[6], S_r, dp[3];
    };
    auto load_level = [&](const VI& rec, Lvl& o) {
      V rw[8], sc[6], td[4];
      ln.template lds_readv<8>(row_group(rec, ix.row6), rw);
      ln.template lds_readv<6>(rec + RL_S, sc);
      ln.template lds_readv<4>(rec + RL_TAU, td);
#pragma unroll
      for (int j = 0; j < 6; ++j) o.Mrow[j] = rw[j], o.S[j] = sc[j];
      o.S_r = rw[RL_ROW_S];
#pragma unroll
      for (int k = 0; k < 3; ++k) o.dp[k] = td[1 + k];
    };
    V accM[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) accM[j] = zero;
    Lvl cur, nxt;
    load_level(rt.rec[kRowLevels - 1], cur);
#pragma unroll
    for (int Lv = kRowLevels - 1; Lv >= 0; --Lv) {
      if (Lv >= 1) load_level(rt.rec[Lv - 1], nxt);
      if (Lv <= max_depth && (Lv >= 1 || floating)) {
        const VM has = rt.rec[Lv] != lds_zero_rec(P.nL);
        V MArow[6];
        if (!L::add6_packed(cur.Mrow, accM, MArow)) {
#pragma unroll
          for (int j = 0; j < 6; ++j) MArow[j] = cur.Mrow[j] + accM[j];
        }
        if (Lv == 0) {
          // the articulated base inertia: its rows meet in the LDS, every lane factorises the same 6x6 (LDL^T) and
          // lane 0 leaves the factor for the main wave's base solve (which then is two substitutions)
          const V rw[8] = {MArow[0], MArow[1], MArow[2], MArow[3], MArow[4], MArow[5], zero, zero};
          ln.template lds_writev_if<8>(ix.row6 * 8 + XB, rw, lane < 6);
          ln.lds_sync();
          V MAs[21];
          const VI z = lane * 0;
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            V r8[8];
            ln.template lds_readv<8>(z + (XB + 8 * i), r8);
#pragma unroll
            for (int j = i; j < 6; ++j) MAs[sidx(i, j)] = r8[j];
          }
          ln.lds_sync();
          TreeFac tf;
          ldl6_factor(MAs, tf, true);
          const V fac[24] = {tf.Lm[0], tf.Lm[1], tf.Lm[2], tf.Lm[3], tf.Lm[4], tf.Lm[5], tf.Lm[6], tf.Lm[7], tf.Lm[8], tf.Lm[9], tf.Lm[10], tf.Lm[11],
                             tf.Lm[12], tf.Lm[13], tf.Lm[14], tf.Di[0], tf.Di[1], tf.Di[2], tf.Di[3], tf.Di[4], tf.Di[5], zero, zero, zero};
          ln.template lds_writev_if<24>(z + XB, fac, lane == 0);
        } else {
          const V S_r = cur.S_r;
          V U[6];
          if (!L::scale6_packed(MArow, S_r, U)) {
#pragma unroll
            for (int j = 0; j < 6; ++j) U[j] = MArow[j] * S_r;
          }
          ln.allreduce8x6(U);
          V U_r, d;
          if (!(L::dot6_packed(MArow, cur.S, &U_r) && L::dot6_packed(cur.S, U, &d))) {
            U_r = MArow[0] * cur.S[0], d = cur.S[0] * U[0];
#pragma unroll
            for (int j = 1; j < 6; ++j) {
              U_r = U_r + MArow[j] * cur.S[j];
              d = d + cur.S[j] * U[j];
            }
          }
          const V inv = vsel(has, vrcp_acc(vsel(has, d, V(T(1)))), zero);
          const V Ud = U_r * inv;
          V Ma[6];
          if (!L::axpy6_packed(MArow, -Ud, U, Ma)) {
#pragma unroll
            for (int j = 0; j < 6; ++j) Ma[j] = MArow[j] - Ud * U[j];
          }
          {
            const V xl[8] = {Ma[0], Ma[1], Ma[2], Ma[3], Ma[4], Ma[5], U_r, inv};
            ln.template lds_writev<8>((lane + Lv * G) * kDuoXlRec + XL, xl);
          }
          if (Lv >= 2 || floating) {
            if (anch && ((cross_levels >> Lv) & 1u)) {
              V d3[3];
#pragma unroll
              for (int k = 0; k < 3; ++k) d3[k] = cur.dp[k];
              {  // row[3:6] -= row[0:3] x d
                const V t0 = Ma[1] * d3[2] - Ma[2] * d3[1], t1 = Ma[2] * d3[0] - Ma[0] * d3[2], t2 = Ma[0] * d3[1] - Ma[1] * d3[0];
                Ma[3] = Ma[3] - t0, Ma[4] = Ma[4] - t1, Ma[5] = Ma[5] - t2;
              }
              const V c1 = vsel(ix.is_ang, pick3(d3, ix.j1), zero);
              const V c2 = vsel(ix.is_ang, -pick3(d3, ix.j2), zero);
              V g1[6], g2[6];
#pragma unroll
              for (int j = 0; j < 6; ++j) g1[j] = ln.shfl(Ma[j], ix.xsrc1), g2[j] = ln.shfl(Ma[j], ix.xsrc2);
              ln.fence();
#pragma unroll
              for (int j = 0; j < 6; ++j) Ma[j] = Ma[j] + c1 * g2[j] + c2 * g1[j];
            }
            const V fcf = vsel(((rt.fcbits >> Lv) & 1) != 0, V(T(1)), zero);
            if (!L::scale6_packed(Ma, fcf, accM)) {
#pragma unroll
              for (int j = 0; j < 6; ++j) accM[j] = Ma[j] * fcf;
            }
            if ((cross_levels >> Lv) & 1u) {
              const int npull = (int)((pull_counts >> (4 * Lv)) & 15u);
#pragma unroll
              for (int k = 0; k < kRowExtra; ++k) {
                if (k >= npull) break;
                const VI src = rt.pull[Lv][k];  // (nothing to pull: an idle row lane, zeros)
                V g[6];
#pragma unroll
                for (int j = 0; j < 6; ++j) g[j] = ln.shfl(Ma[j], src);
                ln.fence();
#pragma unroll
                for (int j = 0; j < 6; ++j) accM[j] = accM[j] + g[j];
              }
            }
          }
        }
        ln.flag_post(kDuoFlagFk + kRowLevels - Lv);  // level Lv (and every deeper one) is in the LDS
      }
      cur = nxt;
      ln.stamp(A, 24 + Lv);
    }
    ln.stamp(A, 7);
  }

  // The main wave's side of passes 2 and 3: bias recursion with the rows of Ma, U_r and 1 / d taken from XL.
  JXS_HD void aba_rows_main(const VI& lane, const RowTabs& rt, const V* pA, const V* S6, const V* c6, const V& tau,
                            const bool anch, const V* dpl, V& sdd, V* a0) const {
    const V zero = V(T(0));
    constexpr int AR = duo_main_off(G);  // this wave's LDS area
    const VI rec_me = own_record(lane) + AR;
    ln.lds_masked(lane < P.nL, [&]() {
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const V ps[2] = {pA[i], S6[i]};
        ln.template lds_writev<2>(rec_me + (RL_ROW + 8 * i + RL_ROW_PA), ps);
      }
      const V sc[12] = {S6[0], S6[1], S6[2], S6[3], S6[4], S6[5], c6[0], c6[1], c6[2], c6[3], c6[4], c6[5]};
      ln.template lds_writev<12>(rec_me + RL_S, sc);
      const V td[4] = {tau, anch ? dpl[0] : zero, anch ? dpl[1] : zero, anch ? dpl[2] : zero};
      ln.template lds_writev<4>(rec_me + RL_TAU, td);
    });
    write_zero_area(lane, AR);
    const int XB = lds_base_rows(G);  // base rows: the inertia wave's area holds the inertia, this area the bias force
    const int XL = duo_xl_off(G);
    RowLaneIdx ix;
    row_lane_idx(lane, ix);
    const int max_depth = P.max_depth;
    const unsigned cross_levels = ln.pin(P.row_cross_levels);
    const unsigned ppull_levels = ln.pin(P.row_ppull_levels);
    const unsigned pull_counts = ln.pin(P.row_pull_counts);
    const int floating = ln.pin(P.floating);
    struct Lvl {
      V pr, S_r, c_r, c[6], tau, dp[3];
    };
    auto load_level = [&](const VI& rec0, Lvl& o) {
      const VI rec = rec0 + AR;
      V ps[2], sc[8], td[4];
      ln.template lds_readv<2>(row_group(rec0, ix.row6) + (AR + RL_ROW_PA), ps);
      ln.template lds_readv<8>(rec + (RL_C - 2), sc);  // {S4, S5, c0..c5}: two aligned 128-bit reads
      ln.template lds_readv<4>(rec + RL_TAU, td);
      o.c_r = ln.lds_read(rec + ix.row6 + RL_C);
      o.pr = ps[0], o.S_r = ps[1];
#pragma unroll
      for (int j = 0; j < 6; ++j) o.c[j] = sc[2 + j];
      o.tau = td[0];
#pragma unroll
      for (int k = 0; k < 3; ++k) o.dp[k] = td[1 + k];
    };
    V Ur[kRowLevels], Sr[kRowLevels], cr[kRowLevels], invd[kRowLevels], uu[kRowLevels];
    V accp = zero, p0 = zero;
    V dkeep[kRowLevels][3];
    Lvl cur, nxt;
    load_level(rt.rec[kRowLevels - 1], cur);
    V xl_next[8];
    bool xl_ahead = false;  // (wave-uniform)
#pragma unroll
    for (int j = 0; j < 8; ++j) xl_next[j] = zero;
#pragma unroll
    for (int Lv = kRowLevels - 1; Lv >= 0; --Lv) {
      Ur[Lv] = zero, Sr[Lv] = zero, cr[Lv] = zero, invd[Lv] = zero, uu[Lv] = zero;
      if (Lv >= 1) load_level(rt.rec[Lv - 1], nxt);
      if (Lv <= max_depth && (Lv >= 1 || floating)) {
        const V pr = cur.pr + accp;
        if (Lv == 0) {
          p0 = pr;
        } else {
          // what the inertia wave handed over for this level: already here when the previous level found it
          // published (the progress word is only read again when the last value seen does not cover the level)
          V xl[8];
          if (xl_ahead) {
#pragma unroll
            for (int j = 0; j < 8; ++j) xl[j] = xl_next[j];
          } else {
            ln.flag_wait(kDuoFlagFk + kRowLevels - Lv, duo_seen_);
            ln.template lds_readv<8>((lane + Lv * G) * kDuoXlRec + XL, xl);
          }
          xl_ahead = false;
          if (Lv >= 2 && duo_seen_ >= kDuoFlagFk + kRowLevels - (Lv - 1)) {
            ln.template lds_readv<8>((lane + (Lv - 1) * G) * kDuoXlRec + XL, xl_next);
            xl_ahead = true;
          }
          const V S_r = cur.S_r;
          const V sp = ln.allreduce8(S_r * pr);
          const V u = cur.tau - sp;
          const V U_r = xl[6], inv = xl[7];
          const V Ud = U_r * inv;
          V pa = pr + Ud * u;
          {
            V t;
            if (L::dot6_packed(xl, cur.c, &t)) {
              pa = pa + t;
            } else {
#pragma unroll
              for (int j = 0; j < 6; ++j) pa = pa + xl[j] * cur.c[j];
            }
          }
          Ur[Lv] = U_r, Sr[Lv] = S_r, cr[Lv] = cur.c_r, invd[Lv] = inv, uu[Lv] = u;
          if (Lv >= 2 || floating) {
            if (anch && ((cross_levels >> Lv) & 1u)) {
              V d3[3];
#pragma unroll
              for (int k = 0; k < 3; ++k) d3[k] = cur.dp[k], dkeep[Lv][k] = d3[k];
              const V c1 = vsel(ix.is_ang, pick3(d3, ix.j1), zero);
              const V c2 = vsel(ix.is_ang, -pick3(d3, ix.j2), zero);
              const V g1 = ln.shfl(pa, ix.xsrc1), g2 = ln.shfl(pa, ix.xsrc2);
              ln.fence();
              pa = pa + c1 * g2 + c2 * g1;
            }
            const VM fc = ((rt.fcbits >> Lv) & 1) != 0;
            accp = vsel(fc, pa, zero);
            if ((cross_levels >> Lv) & 1u) {
              const int npull = (int)((pull_counts >> (4 * Lv)) & 15u);
#pragma unroll
              for (int k = 0; k < kRowExtra; ++k) {
                if (k >= npull) break;
                const VI src = rt.pull[Lv][k];  // (nothing to pull: an idle row lane, zero)
                const V g = ln.shfl(pa, src);
                ln.fence();
                accp = accp + g;
              }
            }
          }
        }
      }
      cur = nxt;
      ln.stamp(A, 24 + Lv);
    }
    ln.stamp(A, 7);  // pass 2 (bias recursion)
    // ---- base acceleration (rbda/aba.py:240-243) --------------------------------------------
    if (floating) {
      ln.lds_write(ix.row6 + (XB + AR), p0, lane < 6);
      ln.lds_sync();
      ln.flag_wait(kDuoFlagFk + kRowLevels, duo_seen_);  // the factor of the articulated base inertia is in the inertia wave's area
      const VI z = lane * 0;
      V fac[24], pAs[8];
      ln.template lds_readv<24>(z + XB, fac);
      ln.template lds_readv<8>(z + (XB + AR), pAs);
      TreeFac tf;
#pragma unroll
      for (int k = 0; k < 15; ++k) tf.Lm[k] = fac[k];
#pragma unroll
      for (int k = 0; k < 6; ++k) tf.Di[k] = fac[15 + k];
      V rhs[6] = {pAs[0], pAs[1], pAs[2], pAs[3], pAs[4], pAs[5]};
      ldl6_solve_neg(tf, rhs, a0);
    } else {
#pragma unroll
      for (int k = 0; k < 6; ++k) a0[k] = zero;
      a0[2] = V(-P.g);
    }
    ln.stamp(A, 8);  // base solve
    // ---- pass 3, base to leaves (rbda/aba.py:251-267): as in aba_rows -----------------------------
    V acar = vsel(ix.row == 0, a0[0], vsel(ix.row == 1, a0[1], vsel(ix.row == 2, a0[2],
             vsel(ix.row == 3, a0[3], vsel(ix.row == 4, a0[4], vsel(ix.row == 5, a0[5], zero))))));
#pragma unroll
    for (int Lv = 1; Lv < kRowLevels; ++Lv) {
      if (Lv <= max_depth) {
        const VM has = rt.rec[Lv] != lds_zero_rec(P.nL);
        V apar = acar;
        if ((ppull_levels >> Lv) & 1u) {
          const VM pulled = rt.ppull[Lv] >= 0;
          const V q = ln.shfl(acar, rt.ppull[Lv]);
          if (anch) {
            const VI pslot = vsel(pulled, rt.ppull[Lv] - ix.row, lane * 0);
            const V al1 = ln.shfl(acar, pslot + 3 + ix.j1), al2 = ln.shfl(acar, pslot + 3 + ix.j2);
            ln.fence();
            const V* d3 = dkeep[Lv];
            const V sh = al1 * pick3(d3, ix.j2) - al2 * pick3(d3, ix.j1);
            apar = vsel(pulled, q + vsel(ix.is_lin, sh, zero), acar);
          } else {
            ln.fence();
            apar = vsel(pulled, q, acar);
          }
        }
        V ai = apar + cr[Lv];
        const V tot = ln.allreduce8(Ur[Lv] * ai);
        const V sd = (uu[Lv] - tot) * invd[Lv];
        ai = ai + Sr[Lv] * sd;
        acar = vsel(has, ai, acar);
        ln.lds_write(rt.rec[Lv] + (RL_SDD + AR), sd);  // (all row lanes: same address, same value; see aba_rows)
      }
    }
    sdd = ln.lds_read(rec_me + RL_SDD);
  }

  template <int OFF>
  JXS_HD void seg_step_dpp(const VI& tail, V* w6) const {
    if (G >= 16) {
      // the partner's value times 1 or 0 -- a 16-lane row holds points of ONE environment here, so a non-finite wrench
      // can only reach lanes of the environment it belongs to
      ln.template fmac6_row_from_next<OFF>(w6, vsel(tail >= OFF, V(T(1)), V(T(0))));
      return;
    }
    const VM take = tail >= OFF;
    V g6[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) g6[k] = ln.template row_from_next<OFF>(w6[k]);
#pragma unroll
    for (int k = 0; k < 6; ++k) w6[k] = w6[k] + vsel(take, g6[k], V(T(0)));
  }

  struct PointSlot {
    VI body, prow, tail, hd, l1;
    V Lp[3], m[3], md[3];  // md: deformation rate of the last evaluation (chunk 0)
  };
  JXS_HD void load_slot_tables(const VI& lane, int ch, PointSlot& ps) const {
    const unsigned char* c = A.chunks + (size_t)ch * chunk_bytes<T>(G);
    const int* pti = reinterpret_cast<const int*>(c);
    const T* ptf = reinterpret_cast<const T*>(c + G * kPtStride * 4);
    const int* head = reinterpret_cast<const int*>(c + G * kPtStride * (4 + (int)sizeof(T)));
    ps.body = ln.ploadi(pti, PI_BODY, lane);
    ps.prow = ln.ploadi(pti, PI_ROW, lane);
    ps.tail = ln.ploadi(pti, PI_TAIL, lane);
    ps.l1 = ln.ploadi(pti, PI_L1, lane);
#pragma unroll
    for (int k = 0; k < 3; ++k) ps.Lp[k] = ln.ploadf(ptf, PF_POS + k, lane);
    ps.hd = ln.hconsti(head, 0);
  }
  JXS_HD void load_slot_state(PointSlot& ps) const {
    // empty slots carry row 0: the load is in range and its value is masked by `valid` later
#pragma unroll
    for (int k = 0; k < 3; ++k) ps.m[k] = ln.gload(A.state_in, ps.prow * 3 + (P.row_m + k), P.n_rows);
  }

#include "jxs_rigid.inc"

  // Per-link sum of the point wrenches (api/contact.py:557-603): segmented suffix-sum over the slots of
  // one link, then every link lane fetches the sum at the head slot of its segment.
  JXS_HD void link_wrench_sums(const VI& lane, const VI& tail, const VI& hd_in, V* w6, V* fl, V* fa) const {
    const V zero = V(T(0));
    // segmented suffix-sum over the slots of one link (slots are sorted by link); when every
    // segment lies inside a 16-lane row the partner lane+off is reached by a DPP row shift
    if (P.seg_dpp_ok) {
      if (P.seg_steps > 0) seg_step_dpp<1>(tail, w6);
      if (P.seg_steps > 1) seg_step_dpp<2>(tail, w6);
      if (P.seg_steps > 2) seg_step_dpp<4>(tail, w6);
      if (P.seg_steps > 3) seg_step_dpp<8>(tail, w6);
    } else {
      for (int st = 0, off = 1; st < P.seg_steps; ++st, off <<= 1) {
        const VM take = tail >= off;
        const VI src = lane + off;
        V g6[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) g6[k] = ln.shfl(w6[k], src);
        ln.fence();
#pragma unroll
        for (int k = 0; k < 6; ++k) w6[k] = w6[k] + vsel(take, g6[k], zero);
      }
    }
    const VI hd = hd_in;
    const VM has = hd >= 0;
    V h6[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) h6[k] = ln.shfl(w6[k], hd);
    ln.fence();
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      fl[k] = fl[k] + vsel(has, h6[k], zero);
      fa[k] = fa[k] + vsel(has, h6[3 + k], zero);
    }
  }

  // ---- terrain (rbda/contacts/common.py:25-63, terrain/terrain.py:15-238) --------------------------------------------
  // [round 6] Height field: the bilinear interpolant of the samples hf[ix * ny + iy] at (x0 + ix dx, y0 + iy dy), clamped to
  // the border samples outside the grid.  Four gathers per evaluation from the model block (L2-resident).
  JXS_HD V hf_height(const V& x, const V& y) const {
    const V zero = V(T(0)), one = V(T(1));
    const V fx = vmin(vmax((x - P.hf_x0) * P.hf_idx, zero), V(T(P.hf_nx - 1)));
    const V fy = vmin(vmax((y - P.hf_y0) * P.hf_idy, zero), V(T(P.hf_ny - 1)));
    // cell index: floor, with the last sample belonging to the last cell (t = 1 there)
    const VI ix = L::to_int(vmin(fx, V(T(P.hf_nx - 2)))), iy = L::to_int(vmin(fy, V(T(P.hf_ny - 2))));
    const V tx = fx - L::to_real(ix), ty = fy - L::to_real(iy);
    const VI b = ix * P.hf_ny + iy;
    const V h00 = ln.tgather(A.hf, b), h01 = ln.tgather(A.hf, b + 1);
    const V h10 = ln.tgather(A.hf, b + P.hf_ny), h11 = ln.tgather(A.hf, b + (P.hf_ny + 1));
    const V a = h00 + (h01 - h00) * ty, c = h10 + (h11 - h10) * ty;
    (void)one;
    return a + (c - a) * tx;
  }
  // Unit normal n at the point's (x, y) and h . n with h = [0, 0, height(x, y) - z] (the penetration depth is its
  // positive part) for the terrains that are not flat: PlaneTerrain (constant normal, terrain.py:127-238) and the height
  // field, whose normal is the reference's central difference of the height function (terrain.py:40-62).
  JXS_HD void terrain_normal_and_depth(const V* pw, V* nh, V& hn) const {
    if (P.hf) {
      const V d = V(P.hf_delta);
      const V hxp = hf_height(pw[0] + d, pw[1]), hxm = hf_height(pw[0] - d, pw[1]);
      const V hyp = hf_height(pw[0], pw[1] + d), hym = hf_height(pw[0], pw[1] - d);
      const V nx = (hxm - hxp) * P.hf_inv_2delta, ny = (hym - hyp) * P.hf_inv_2delta;
      const V inv = vrsqrt(nx * nx + ny * ny + V(T(1)));
      nh[0] = nx * inv, nh[1] = ny * inv, nh[2] = inv;
      hn = (hf_height(pw[0], pw[1]) - pw[2]) * nh[2];
    } else {
      nh[0] = V(P.nrm[0]), nh[1] = V(P.nrm[1]), nh[2] = V(P.nrm[2]);
      const V height = V(P.terrain_h) - (P.nrm[0] * pw[0] + P.nrm[1] * pw[1]) * (T(1) / P.nrm[2]);
      hn = (height - pw[2]) * P.nrm[2];
    }
  }

  // Kinematics, penetration and Hunt-Crossley force of one collidable point per lane, given the
  // kinematics (Rb, rb, vbl, vba) of its parent link in frame C: the wrench w6 in C and the rate md of
  // the tangential deformation (rbda/collidable_points.py:9-65, rbda/contacts/common.py:25-63,
  // rbda/contacts/soft.py:195-388).
  JXS_HD void point_physics(const VM& valid, const V* Lp, const V* m, const V* Rb, const V* rb, const V* vbl,
                            const V* vba, const V* rab, const V* pB, const V* doff, const V* vBc, const V* om, V* w6, V* md) const {
    const V zero = V(T(0));
    V rc0[3], rc[3], pw[3], pd[3], t[3];
    mat3vec(Rb, Lp, rc0);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      rc0[k] = rc0[k] + rb[k];                               // relative to the ABA base origin
      rc[k] = P.has_base_off ? rc0[k] + doff[k] : rc0[k];    // relative to the base position, cached (FK) placement
      pw[k] = rc[k] + pB[k];                                 // world position
    }
    // pdot_C = W_v_L,lin + W_w_L x W_p_C of the cached link kinematics
    // (collidable_points.py:50-53), written about the C origin.
    cross(vba, rc0, t);
#pragma unroll
    for (int k = 0; k < 3; ++k) pd[k] = vbl[k] + t[k];
    if (!P.floating) {
      // cached link velocities of a fixed-base model include the stored base velocity
      cross(om, rc, t);
#pragma unroll
      for (int k = 0; k < 3; ++k) pd[k] = pd[k] + vBc[k] + t[k];
    } else if (P.has_base_off) {
      cross(om, doff, t);  // zero for URDF models (suc_H_i[0] = I when floating)
#pragma unroll
      for (int k = 0; k < 3; ++k) pd[k] = pd[k] + t[k];
    }
    // penetration data (rbda/contacts/common.py:25-63): h = [0,0,height(x,y) - p_z],
    // delta = max(0, h.n); FlatTerrain: n = +z; PlaneTerrain: constant unit normal,
    // height(x,y) = h0 - (A x + B y) / C  (terrain/terrain.py:180-215)
    V delta, pdn, mdn;  // penetration, pdot.n, m.n
    V nh[3];
    if (P.flat) {
      nh[0] = zero, nh[1] = zero, nh[2] = V(T(1));
      delta = vmax(zero, V(P.terrain_h) - pw[2]);
      pdn = pd[2];
      mdn = m[2];
    } else {
      V hn;
      terrain_normal_and_depth(pw, nh, hn);
      delta = vmax(zero, hn);
      pdn = pd[0] * nh[0] + pd[1] * nh[1] + pd[2] * nh[2];
      mdn = m[0] * nh[0] + m[1] * nh[1] + m[2] * nh[2];
    }
    const VM in_contact = delta > zero;
    const V ddelta = vsel(in_contact, -pdn, zero);
    V dp, dq;
    if (P.pq_half) {
      dp = vsqrt(delta + P.eps);
      dq = dp;
    } else {
      dp = vpow(delta + P.eps, V(P.p));
      dq = vpow(delta + P.eps, V(P.q));
    }
    const V Kdp = P.K * dp, Ddq = P.D * dq;
    const V fn = vmax(zero, Kdp * delta + Ddq * ddelta);
    // tangential / normal split of the point velocity and of the deformation
    V vt[3], mn[3], mt[3];
    if (P.flat) {
      // n = +z: the products with the components 0, 0, 1 of the normal written out (IEEE arithmetic keeps x * 0 and
      // x - x * 1 as instructions): the same values, signed zeros aside
      vt[0] = pd[0], vt[1] = pd[1], vt[2] = zero;
      mn[0] = zero, mn[1] = zero, mn[2] = m[2];
      mt[0] = m[0], mt[1] = m[1], mt[2] = zero;
    } else {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        vt[k] = pd[k] - pdn * nh[k];
        mn[k] = mdn * nh[k];
        mt[k] = m[k] - mn[k];
      }
    }
    V ft[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) ft[k] = -(Kdp * mt[k] + Ddq * vt[k]);
    if (P.flat) ft[2] = zero;
    const V ft2 = P.flat ? ft[0] * ft[0] + ft[1] * ft[1] : ft[0] * ft[0] + ft[1] * ft[1] + ft[2] * ft[2];
    const V mufn = P.mu * fn;
    const VM no_contact = !in_contact;  // delta <= 0
    const VM sticking = no_contact || (ft2 <= mufn * mufn);
    const V nrm = vsqrt(ft2);
    const V scale = vmin(mufn, nrm) * vrcp(nrm + vsel(nrm == zero, V(P.eps), zero));
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      ft[k] = vsel(sticking, ft[k], scale * ft[k]);
      ft[k] = vsel(no_contact, zero, ft[k]);
    }
    // deformation rate: no contact | sticking | slipping
    const V inv_Ddq = vrcp(Ddq);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const V md_nc = -(P.K_over_D * m[k]);
      if (P.flat && k < 2) {  // (m_n = 0 in the plane)
        const V md_sl = -(ft[k] + Kdp * mt[k]) * inv_Ddq;
        md[k] = vsel(no_contact, md_nc, vsel(sticking, vt[k], md_sl));
      } else if (P.flat) {    // (along the normal: v_t = m_t = f_t = 0)
        md[k] = vsel(no_contact || sticking, md_nc, zero);
      } else {
        const V md_st = vt[k] - P.K_over_D * mn[k];
        const V md_sl = -(ft[k] + Kdp * mt[k]) * inv_Ddq;
        md[k] = vsel(no_contact, md_nc, vsel(sticking, md_st, md_sl));
      }
    }
    // wrench [f; (r_C - rab) x f]  (W_f = [f; p x f], soft.py:377-388, moved to the reference point)
    if (P.flat) {
      w6[0] = vsel(valid, ft[0], zero), w6[1] = vsel(valid, ft[1], zero), w6[2] = vsel(valid, fn, zero);
    } else {
      w6[0] = vsel(valid, fn * nh[0] + ft[0], zero);
      w6[1] = vsel(valid, fn * nh[1] + ft[1], zero);
      w6[2] = vsel(valid, fn * nh[2] + ft[2], zero);
    }
    // moment about the anchor of the parent link's chain (rab = 0: about the origin of C)
    V lever[3] = {rc[0] - rab[0], rc[1] - rab[1], rc[2] - rab[2]};
    cross(lever, w6, w6 + 3);
  }

  // One chunk of G collidable points.  Chunk 0 is carried in registers (state loaded up front, integrated
  // by run()); further chunks go through memory every step (rollouts are not fused then).  Two separate
  // instantiations instead of one loop that selects between `ps0` and a local slot: the select would be a
  // pointer phi that keeps the slot structs in scratch memory.
  // [round 4] kStages (RungeKutta4, chunks behind the first): ps.m is the deformation of the state the step starts
  // from (read from memory at every stage); the stage's deformation is m0 + h * (rate of the previous stage), the new
  // state m0 + dt/6 * (k1 + 2 k2 + 2 k3 + k4) is stored at the last stage (api/integrators.py:91-167).  The rate of
  // the previous stage and the weighted sum live in the LDS, eight words per slot (`sc`: word offset of this lane's
  // slot), touched by the slot's own lane only -- no synchronisation beyond program order.
  template <bool kFirst, bool kStages = false, bool kRates = false>
  JXS_HD void contact_chunk(const VI& lane, PointSlot& ps, const V* R, const V* r, const V* vl, const V* va,
                            const V* ra, const V* pB, const V* doff, const V* vBc, const V* om, V* fl, V* fa,
                            int stage = 0, const VI* sc = nullptr) const {
    const V zero = V(T(0));
    const VM valid = ps.body >= 0;
    V m[3], m0[3], ksum[3] = {zero, zero, zero};
#pragma unroll
    for (int k = 0; k < 3; ++k) m[k] = m0[k] = vsel(valid, ps.m[k], zero);
    if (kStages && stage > 0) {
      V s6[kRk4SlotWords];
      ln.template lds_readv<kRk4SlotWords>(*sc, s6);
      const V h = V(stage == 3 ? P.dt : P.dt * T(0.5));  // euler_mid, euler_mid, euler_fin
#pragma unroll
      for (int k = 0; k < 3; ++k) m[k] = m0[k] + h * s6[k], ksum[k] = s6[3 + k];
    }
    // kinematics of the parent link
    V Rb[9], rb[3], vbl[3], vba[3], rab[3];
#pragma unroll
    for (int e = 0; e < 9; ++e) Rb[e] = ln.shfl(R[e], ps.body);
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      rb[e] = ln.shfl(r[e], ps.body);
      vbl[e] = ln.shfl(vl[e], ps.body);
      vba[e] = ln.shfl(va[e], ps.body);
      rab[e] = ln.shfl(ra[e], ps.body);
    }
    ln.fence();
    V w6[6], md[3];
    point_physics(valid, ps.Lp, m, Rb, rb, vbl, vba, rab, pB, doff, vBc, om, w6, md);
    if (kStages) {
      const T wgt = (stage == 0 || stage == 3) ? T(1) : T(2);
      V s6[kRk4SlotWords];
#pragma unroll
      for (int k = 0; k < 3; ++k) s6[k] = md[k], s6[3 + k] = ksum[k] + wgt * md[k];
      s6[6] = s6[7] = zero;
      if (stage < 3) {
        ln.template lds_writev<kRk4SlotWords>(*sc, s6);
      } else {
#pragma unroll
        for (int k = 0; k < 3; ++k)
          ln.gstore(A.state_out, ps.prow * 3 + (P.row_m + k), m0[k] + (P.dt * T(1.0 / 6.0)) * s6[3 + k], valid, P.n_rows);
      }
    } else {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (kFirst) ps.md[k] = md[k];
      else if (kRates) {  // MODE_DYN: the rate itself is the output (row of m -> mdot), where the derivative block is asked for
        if (A.state_out != nullptr) ln.gstore(A.state_out, ps.prow * 3 + (P.row_m + k), md[k], valid, P.n_rows);
      } else ln.gstore(A.state_out, ps.prow * 3 + (P.row_m + k), m[k] + P.dt * md[k], valid, P.n_rows);
    }
    }
    link_wrench_sums(lane, ps.tail, ps.hd, w6, fl, fa);
  }

  template <bool kStages = false, bool kRates = false>
  JXS_HD void contacts(const VI& lane, PointSlot& ps0, const V* R, const V* r, const V* vl, const V* va,
                       const V* ra, const V* pB, const V* doff, const V* vBc, const V* om, V* fl, V* fa, int stage = 0) const {
    contact_chunk<true>(lane, ps0, R, r, vl, va, ra, pB, doff, vBc, om, fl, fa);  // sets ps0.md
    for (int ch = 1; ch < P.n_chunks; ++ch) {
      PointSlot ps;
      load_slot_tables(lane, ch, ps);
      load_slot_state(ps);
      const VI sc = (lane + (ch - 1) * G) * kRk4SlotWords + rk4_chunk_off(G);
      contact_chunk<false, kStages, kRates>(lane, ps, R, r, vl, va, ra, pB, doff, vBc, om, fl, fa, stage, &sc);
    }
  }

  // ==========================================================================================
  // Joint torques of the gravity term g(q) = RNEA(q, v = 0, vdot = 0) (api/model.py:1897-1931, rbda/rnea.py:12-238).
  // With zero velocities and accelerations every link has the spatial acceleration a_0 = -W_g of the base
  // (rnea.py:139-152: a_i = a_lambda + S sdd + v x vJ), so f_i = M_i a_0 = [m_i a ; c_i x m_i a] with a = (0, 0, -g)
  // in the world-aligned frame C: three non-zero components (f_z, n_x, n_y).  Backward pass (rnea.py:193-219) =
  // subtree sums, tau_i = S_i . f_i.  MODE_GRAV: no velocity rows, no prefix sums, no rotated inertias.
  JXS_HD void gravity_torques(const VI& lane, const VI& jrow, const VI& level, const VI* child, const VM& is_joint,
                              const V* R, const V* r, const V* cL, const V& mass, const V* Sl, const V* Sa) const {
    const V tq = gravity_tq(level, child, R, r, cL, mass, Sl, Sa);
    if (A.out_tau != nullptr) ln.gstore(A.out_tau, jrow, tq, is_joint, P.n);
    if (A.out_a != nullptr) ln.gstore(A.out_a, jrow + 6, tq, is_joint, 6 + P.n);
  }
  // (the value per joint lane: also used by the gravity-compensated step of the rigid contact modes, KArgs::flags bit 1)
  JXS_HD V gravity_tq(const VI& level, const VI* child, const V* R, const V* r, const V* cL, const V& mass, const V* Sl,
                      const V* Sa) const {
    const V zero = V(T(0));
    V cw[3];
    mat3vec(R, cL, cw);
    const V fz = mass * V(-P.g);  // (padding lanes carry mass 0)
    V f3[3] = {fz, (cw[1] + r[1]) * fz, -((cw[0] + r[0]) * fz)};  // f_z, n_x = c_y f_z, n_y = -c_x f_z
    const int first_level = P.floating ? 1 : 2;
    for (int Lv = P.max_depth; Lv >= first_level; --Lv) {
      const VM is_par = level == (Lv - 1);
      const int nch = P.maxch(Lv);
      if (nch >= 1) {
        const VM ok = is_par && (child[0] >= 0);
#pragma unroll
        for (int e = 0; e < 3; ++e) f3[e] = f3[e] + vsel(ok, ln.from_next(f3[e]), zero);
      }
#pragma unroll
      for (int k = 1; k < kMaxChildren; ++k) {
        if (k >= P.max_children) break;  // (the widest link of the MODEL: a constant of a model-specialised kernel -- no copies, no child registers beyond it; the level's own width is tested below)
        if (k < nch) {
          const VM ok = is_par && (child[k] >= 0);
          V g3[3];
#pragma unroll
          for (int e = 0; e < 3; ++e) g3[e] = ln.shfl(f3[e], child[k]);
          ln.fence();
#pragma unroll
          for (int e = 0; e < 3; ++e) f3[e] = f3[e] + vsel(ok, g3[e], zero);
        }
      }
    }
    return Sl[2] * f3[0] + Sa[0] * f3[1] + Sa[1] * f3[2];
  }

  // ==========================================================================================
  // R: RNEA in frame C (rbda/rnea.py:12-238).  in_a = inertial base acceleration + sdd.
  JXS_HD void rnea(const VI& lane, const VI& jrow, const VI& level, const VI* jump, const VI* child,
                   const VM& is_joint, const VM& is_root, const V* Sl, const V* Sa, const V* cl, const V* ca,
                   const V& mass, const V* cw, const V* Ic, const V* bl, const V* ba, const V* fl, const V* fa,
                   const V* pB) const {
    const V zero = V(T(0));
    const VI jrow_c = vsel(jrow >= 0, jrow, lane * 0);
    const V sdd = (A.in_a != nullptr) ? vsel(is_joint, ln.gload(A.in_a, jrow_c + 6, 6 + P.n), zero) : zero;
    // base acceleration in C: a_0 = (Wdot_v - W_g) moved to the C origin (floating) or -W_g
    V al[3], aa[3];
    {
      V wl[3] = {zero, zero, zero}, wa[3] = {zero, zero, zero};
      if (P.floating && A.in_a != nullptr) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          wl[k] = ln.gload_u(A.in_a, k, 6 + P.n);
          wa[k] = ln.gload_u(A.in_a, 3 + k, 6 + P.n);
        }
      }
      V t[3];
      cross(wa, pB, t);  // a_lin^C = a_lin^W + wdot x p_B
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const V base_l = wl[k] + t[k], base_a = wa[k];
        al[k] = vsel(is_root, base_l, Sl[k] * sdd + cl[k]);
        aa[k] = vsel(is_root, base_a, Sa[k] * sdd + ca[k]);
      }
      al[2] = al[2] - vsel(is_root, V(P.g), zero);
    }
    prefix6(jump, al, aa);  // a_i = a_lambda + S sdd + v x vJ  (rnea.py:150-152)
    // f_i = M a + v x* M v - f_ext  (rnea.py:163-168)
    V f6[6];
    {
      V t[3], Ml[3], Ma_[3];
      cross(aa, cw, t);
#pragma unroll
      for (int k = 0; k < 3; ++k) Ml[k] = mass * (al[k] + t[k]);
      V Iw[3];
      Iw[0] = Ic[0] * aa[0] + Ic[1] * aa[1] + Ic[2] * aa[2];
      Iw[1] = Ic[1] * aa[0] + Ic[3] * aa[1] + Ic[4] * aa[2];
      Iw[2] = Ic[2] * aa[0] + Ic[4] * aa[1] + Ic[5] * aa[2];
      cross(cw, Ml, t);
#pragma unroll
      for (int k = 0; k < 3; ++k) Ma_[k] = Iw[k] + t[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        f6[k] = Ml[k] + bl[k] - fl[k];
        f6[3 + k] = Ma_[k] + ba[k] - fa[k];
      }
      // The reference leaves f_0 = 0 for a fixed base (rnea.py:114-131).
      if (!P.floating) {
#pragma unroll
        for (int k = 0; k < 6; ++k) f6[k] = vsel(is_root, zero, f6[k]);
      }
    }
    // backward pass: f_lambda += f_i, one level per iteration (rnea.py:193-219)
    const int first_level = P.floating ? 1 : 2;
    for (int Lv = P.max_depth; Lv >= first_level; --Lv) {
      const VM is_par = level == (Lv - 1);
      const int nch = P.maxch(Lv);
      // children push only when they are at the current level (f of deeper lanes is final, but a
      // parent must not re-add a child it gathered in an earlier iteration)
      if (nch >= 1) {
        const VM ok = is_par && (child[0] >= 0);
#pragma unroll
        for (int e = 0; e < 6; ++e) f6[e] = f6[e] + vsel(ok, ln.from_next(f6[e]), zero);
      }
#pragma unroll
      for (int k = 1; k < kMaxChildren; ++k) {
        if (k >= P.max_children) break;  // (the widest link of the MODEL: a constant of a model-specialised kernel -- no copies, no child registers beyond it; the level's own width is tested below)
        if (k < nch) {
          const VM ok = is_par && (child[k] >= 0);
          V g6[6];
#pragma unroll
          for (int e = 0; e < 6; ++e) g6[e] = ln.shfl(f6[e], child[k]);
          ln.fence();
#pragma unroll
          for (int e = 0; e < 6; ++e) f6[e] = f6[e] + vsel(ok, g6[e], zero);
        }
      }
    }
    V tq = Sl[0] * f6[0] + Sl[1] * f6[1] + Sl[2] * f6[2] + Sa[0] * f6[3] + Sa[1] * f6[4] + Sa[2] * f6[5];
    if (A.out_tau != nullptr) ln.gstore(A.out_tau, jrow, tq, is_joint, P.n);
    if (A.out_a == nullptr) return;
    ln.gstore(A.out_a, jrow + 6, tq, is_joint, 6 + P.n);
    // W_f0 = B_X_W^T f_0: move the base wrench from the C origin back to the world origin
    V t[3];
    cross(pB, f6, t);
    const VI zl = lane * 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      ln.gstore(A.out_a, zl + k, f6[k], is_root, 6 + P.n);
      ln.gstore(A.out_a, zl + (3 + k), f6[3 + k] + t[k], is_root, 6 + P.n);
    }
  }

  // ==========================================================================================
  // Composite-rigid-body algorithm (rbda/crba.py:10-170) in frame C.  The composite inertias are plain
  // subtree sums (all neighbour transforms are the identity), F_i = Ic_i S_i, and
  //     M[6+i, 6+j] = S_j . F_i   for every ancestor-or-self j of i,   M[0:6, 6+i] = F_i,   M[0:6, 0:6] = Ic_0.
  // Frame C = origin at the base position with world-aligned axes, which IS the Mixed velocity representation
  // of the base (api/common.py:39-47): the kernel writes M in Mixed representation, the host applies the 6x6
  // congruence of api/model.py:1529-1590 (`_transform_M_block`) for Body / Inertial.  The output
  // out_a[(6+n)^2][N] is zeroed by the caller; only the structurally non-zero entries are written.
  JXS_HD void crba(const VI& lane, const VI& level, const VI* child, const VI& jrow, const VM& is_joint,
                   const VM& is_root, const V* M_link, const V* S6) const {
    V Ic[21];
#pragma unroll
    for (int e = 0; e < 21; ++e) Ic[e] = M_link[e];
    // composite inertias, leaves to base: parents at level Lv-1 add their children (all at level Lv, final)
    for (int Lv = P.max_depth; Lv >= 1; --Lv) {
      const VM is_par = level == (Lv - 1);
      const int nch = P.maxch(Lv);
      if (nch >= 1) {
        const VM ok0 = is_par && (child[0] >= 0);
        V t9[9];
        add_from_next<9>(lane, Ic, Ic, ok0);
        add_from_next<9>(lane, Ic + 9, Ic + 9, ok0);
#pragma unroll
        for (int e = 0; e < 3; ++e) t9[e] = Ic[18 + e];
#pragma unroll
        for (int e = 3; e < 9; ++e) t9[e] = V(T(0));
        V a9[9];
#pragma unroll
        for (int e = 0; e < 9; ++e) a9[e] = t9[e];
        add_from_next<9>(lane, a9, t9, ok0);
#pragma unroll
        for (int e = 0; e < 3; ++e) Ic[18 + e] = a9[e];
      }
#pragma unroll
      for (int k = 1; k < kMaxChildren; ++k) {
        if (k >= P.max_children) break;  // (the widest link of the MODEL: a constant of a model-specialised kernel -- no copies, no child registers beyond it; the level's own width is tested below)
        if (k < nch) {
          const V okf = vsel(is_par && (child[k] >= 0), V(T(1)), V(T(0)));
          V g[21];
#pragma unroll
          for (int e = 0; e < 21; ++e) g[e] = ln.shfl(Ic[e], child[k]);
          ln.fence();
#pragma unroll
          for (int e = 0; e < 21; ++e) Ic[e] = Ic[e] + okf * g[e];
        }
      }
    }
    V F[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      V acc = Ic[sidx(i, 0)] * S6[0];
#pragma unroll
      for (int j = 1; j < 6; ++j) acc = acc + Ic[sidx(i, j)] * S6[j];
      F[i] = acc;
    }
    const int nv = 6 + P.n, rows = nv * nv;
    const VI sub = ln.lconsti(A.lti, LI_SUBTREE);
    // base block: the composite inertia of the whole tree
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) ln.gstore(A.out_a, lane * 0 + (i * nv + j), Ic[sidx(i, j)], is_root, rows);
    // one pass per link lane k: every lane that is an ancestor-or-self joint of k writes its entry (and the
    // mirrored one), the base lane writes the coupling column F_k
    for (int k = 1; k < P.nL; ++k) {
      const VI src = lane * 0 + k;
      V Fk[6];
#pragma unroll
      for (int e = 0; e < 6; ++e) Fk[e] = ln.shfl(F[e], src);
      const VI ck = ln.shfl(jrow, src) + 6;  // column of joint k
      ln.fence();
      V v = S6[0] * Fk[0];
#pragma unroll
      for (int e = 1; e < 6; ++e) v = v + S6[e] * Fk[e];
      const VM anc = is_joint && (lane <= src) && (src < lane + sub);
      const VI rj = jrow + 6;
      ln.gstore(A.out_a, rj * nv + ck, v, anc, rows);
      ln.gstore(A.out_a, ck * nv + rj, v, anc, rows);
#pragma unroll
      for (int e = 0; e < 6; ++e) {
        ln.gstore(A.out_a, ck + e * nv, Fk[e], is_root, rows);
        ln.gstore(A.out_a, ck * nv + e, Fk[e], is_root, rows);
      }
    }
  }

  // ==========================================================================================
  // Inverse of the free-floating mass matrix (rbda/mass_inverse.py:11-233; api/model.py:1593-1631).  The
  // reference runs a dedicated propagation; here column c of M^-1 is the response of the articulated-body
  // factorisation (pass 2 above: U, 1/d, the LDL^T of the articulated base inertia) to the unit generalized
  // force e_c -- a unit wrench on the base link for c < 6, a unit joint torque otherwise -- three columns per
  // sweep.  Like the mass-matrix kernel the result is in MIXED representation (frame C); out_a = [(6+n)^2][N].
  // As in the reference the base link is a free body for EVERY model -- fixed-base ones too -- so the result is
  // the inverse of the full (6+n) free-floating mass matrix and its joint block the Schur-complement inverse.
  JXS_HD void mass_inverse(const VI& lane, const VI& level, const VI& parent, const VI* child, const VI& jrow,
                           const VM& is_joint, const VM& is_root, const V* MA, const V* U, const V* S6,
                           const V& inv_d) const {
    TreeFac tf;
#pragma unroll
    for (int k = 0; k < 6; ++k) tf.U[k] = U[k], tf.S6[k] = S6[k];
    tf.inv_d = inv_d;
    ldl6_factor(MA, tf);  // the base is a free 6-DoF body for every model here (D0 = I_A[0], mass_inverse.py:160-166)
    const int nv = 6 + P.n, rows = nv * nv;
    const VI zl = lane * 0;
    const V zero = V(T(0)), one = V(T(1));
    for (int c0 = 0; c0 < nv; c0 += 3) {
      V pAr[3][6], ar[3][6], sddr[3], taur[3];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int c = c0 + r;  // wave-uniform column index
#pragma unroll
        for (int i = 0; i < 6; ++i) pAr[r][i] = vsel(is_root && (zl + c == i), -one, zero);  // pA = -wrench
        taur[r] = vsel(is_joint && (jrow + 6 == zl + c), one, zero);
      }
      response<3>(lane, level, parent, child, tf, pAr, ar, sddr, taur, 1);
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int c = c0 + r;
        if (c < nv) {
#pragma unroll
          for (int i = 0; i < 6; ++i) ln.gstore(A.out_a, zl + (i * nv + c), ar[r][i], is_root, rows);
          ln.gstore(A.out_a, (jrow + 6) * nv + c, sddr[r], is_joint, rows);
        }
      }
    }
  }

  // ==========================================================================================
  // Doubly-left full Jacobian `B_J_full_WX_B` [6][6+n], its derivative `B_Jdot_full_WX_B` and the link
  // poses `B_H_L` relative to the base (rbda/jacobian.py:128-339): everything the Jacobian API of the
  // reference is assembled from (api/model.py:925-1228, api/contact.py:214-511).  Column 6+j of the full
  // Jacobian is B_X_j S_j = the motion subspace of joint j in the base frame, the first six columns are the
  // identity; column 6+j of the derivative is  B_v_{B,j} x (B_X_j S_j)  with the velocity of link j RELATIVE to
  // the base (joint velocities only).  Frame C has its origin at the base position, so a rotation by R0^T is
  // all that separates it from the base frame B.  out_a = [2*6*(6+n)][N] (J then Jdot, zeroed by the caller),
  // out_H = [nL*12][N].
  JXS_HD void jacobians(const VI& lane, const VI& lnk, const VI& level, const VI& jrow, const VM& is_joint,
                        const VM& is_root, const V* R0, const V* R, const V* r, const V* Sl, const V* Sa,
                        const V* vl, const V* va, const V* vBc, const V* om) const {
    const VI zl = lane * 0;
    const int nv = 6 + P.n, rows = 12 * nv;
    // velocity field of the base (the link velocities of a fixed-base model start from zero, see run())
    V v0l[3], v0a[3];
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      v0l[e] = P.floating ? vBc[e] : V(T(0));
      v0a[e] = P.floating ? om[e] : V(T(0));
    }
    auto to_base = [&](const V* x, V* o) {  // o = R0^T x
      o[0] = R0[0] * x[0] + R0[3] * x[1] + R0[6] * x[2];
      o[1] = R0[1] * x[0] + R0[4] * x[1] + R0[7] * x[2];
      o[2] = R0[2] * x[0] + R0[5] * x[1] + R0[8] * x[2];
    };
    V sl[3], sa[3], wl[3], wa[3], t[3];
    to_base(Sl, sl);
    to_base(Sa, sa);
#pragma unroll
    for (int e = 0; e < 3; ++e) t[e] = vl[e] - v0l[e];
    to_base(t, wl);
#pragma unroll
    for (int e = 0; e < 3; ++e) t[e] = va[e] - v0a[e];
    to_base(t, wa);
    // Jdot column = crm(w) s = [wa x sl + wl x sa ; wa x sa]   (math/cross.py:14-43)
    V dl[3], da[3], t0[3], t1[3];
    cross(wa, sl, t0);
    cross(wl, sa, t1);
    cross(wa, sa, da);
#pragma unroll
    for (int e = 0; e < 3; ++e) dl[e] = t0[e] + t1[e];
    const VI col = jrow + 6;
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      ln.gstore(A.out_a, col + e * nv, sl[e], is_joint, rows);
      ln.gstore(A.out_a, col + (3 + e) * nv, sa[e], is_joint, rows);
      ln.gstore(A.out_a, col + (6 + e) * nv, dl[e], is_joint, rows);
      ln.gstore(A.out_a, col + (9 + e) * nv, da[e], is_joint, rows);
    }
#pragma unroll
    for (int e = 0; e < 6; ++e) ln.gstore(A.out_a, zl + (e * nv + e), V(T(1)), is_root, rows);
    if (A.out_H != nullptr) {
      // B_H_L = [R0^T R | R0^T r]
      const VM is_link = level >= 0;
      V c0[3] = {R[0], R[3], R[6]}, c1[3] = {R[1], R[4], R[7]}, c2[3] = {R[2], R[5], R[8]}, b0[3], b1[3], b2[3], bp[3];
      to_base(c0, b0);
      to_base(c1, b1);
      to_base(c2, b2);
      to_base(r, bp);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        ln.gstore(A.out_H, lnk * 12 + (4 * i + 0), b0[i], is_link, P.nL * 12);
        ln.gstore(A.out_H, lnk * 12 + (4 * i + 1), b1[i], is_link, P.nL * 12);
        ln.gstore(A.out_H, lnk * 12 + (4 * i + 2), b2[i], is_link, P.nL * 12);
        ln.gstore(A.out_H, lnk * 12 + (4 * i + 3), bp[i], is_link, P.nL * 12);
      }
    }
  }

  // ==========================================================================================
  // N: cached kinematics -- W_H_L and inertial-fixed W_v_WL of every link (api/data.py:480-492)
  JXS_HD void store_kinematics(const VI& lnk, const VI& level, const V* R, const V* r, const V* vl,
                               const V* va, const V* pB, const V* doff, const V* vBc, const V* om) const {
    const VM is_link = level >= 0;
    V p[3], vlin[3], vang[3], t[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      p[k] = r[k] + doff[k] + pB[k];
      vlin[k] = vl[k];
      vang[k] = va[k];
    }
    if (!P.floating) {
      // fixed base: the cache starts from the stored base velocity (forward_kinematics.py:69-70)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        vlin[k] = vlin[k] + vBc[k];
        vang[k] = vang[k] + om[k];
      }
    }
    // spatial velocity about the C origin -> about the world origin: v_lin^W = v_lin^C - w x p_B
    // (the cached transforms are offset by doff, whose effect on W_X_i S is w_i x doff)
    cross(vang, pB, t);
#pragma unroll
    for (int k = 0; k < 3; ++k) vlin[k] = vlin[k] - t[k];
    {
      // linear part uses the cached positions (r + doff): add (r+doff) x S_ang contributions,
      // i.e. doff x (w_i - w_base) for the joint part.
      V wrel[3], t2[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) wrel[k] = va[k] - (P.floating ? om[k] : V(T(0)));
      cross(doff, wrel, t2);
#pragma unroll
      for (int k = 0; k < 3; ++k) vlin[k] = vlin[k] + t2[k];
    }
    if (A.out_H != nullptr) {
#pragma unroll
      for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) ln.gstore(A.out_H, lnk * 12 + (4 * i + j), R[3 * i + j], is_link, P.nL * 12);
        ln.gstore(A.out_H, lnk * 12 + (4 * i + 3), p[i], is_link, P.nL * 12);
      }
    }
    if (A.out_V != nullptr) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        ln.gstore(A.out_V, lnk * 6 + k, vlin[k], is_link, P.nL * 6);
        ln.gstore(A.out_V, lnk * 6 + (3 + k), vang[k], is_link, P.nL * 6);
      }
    }
  }
};

}  // namespace jxs
