// Kernel parameter block + per-lane table layout shared by the host packer, the HIP
// kernels and the host lockstep emulation used by the CPU tests.
//
// Mapping: one environment is processed by a group of G lanes of a
// wavefront (G = 4..64, power of two).  Links are assigned to lanes in depth-first pre-order
// (NOT the reference's BFS link index, `src/jaxsim/parsers/kinematic_graph.py:133-134`): the
// first child of the link in lane j sits in lane j+1, so the most frequent parent<->child
// exchanges of the ABA sweeps are DPP lane shifts (VALU modifiers) instead of ds_bpermute
// round trips (~190 cycles each, measured).  The per-lane tables carry the reference link /
// joint index of every lane.  In the contact phase lane j owns point slot j of the chunk.
#pragma once
#include <cstdint>

#ifndef JXS_HD
#if defined(__HIPCC__)
#define JXS_HD __host__ __device__ __forceinline__
#else
#define JXS_HD inline
#endif
#endif

namespace jxs {

constexpr int kMaxDepth = 63;    // max tree depth supported by the level loops
constexpr int kMaxChildren = 12;  // max children of one link (statically unrolled gathers that stop at the widest link of the level; [round 6] was 6)
constexpr int kMaxRounds = 6;    // pointer-jumping rounds: ceil(log2(depth+1)) <= 6

// ---- per-lane tables are LANE-MAJOR: tbl[lane * stride + field], strides multiples of 4 words and the
// tables 16-byte aligned, so that one lane's record is read with a few wide loads (global_load_dwordx4)
// instead of one dword load per field (round 1: ~110 table loads per wave, each costing a lone wave its
// issue slot + address arithmetic; now ~30).  Both environments of a wave read the same addresses.
// ---- per-lane float table fields: ltf[lane * kLtfStride + field] ---------------------------
enum LaneF : int {
  LF_RPRE = 0,    // 9: rotation of lambda_H_pre (row-major)   math/joint_model.py:70-98
  LF_PPRE = 9,    // 3: translation of lambda_H_pre
  LF_AXIS = 12,   // 3: unit joint axis
  LF_MASS = 15,   // 1
  LF_COM = 16,    // 3: CoM in the link frame
  LF_ICOM = 19,   // 6: I_CoM xx,xy,xz,yy,yz,zz
  LF_KC = 25,     // friction_static       kin_dyn_parameters.py:502-571
  LF_KV = 26,     // friction_viscous
  LF_SMIN = 27,   // position_limits_min
  LF_SMAX = 28,   // position_limits_max
  LF_KLIM = 29,   // position_limit_spring
  LF_DLIM = 30,   // position_limit_damper
  LF_RSUC = 32,   // 9: rotation of suc_H_i (read only when some suc_H_i is not the identity)
  LF_PSUC = 41,   // 3: translation of suc_H_i
  LF_COUNT = 44
};
constexpr int kLtfStride = LF_COUNT;

// ---- per-lane int table fields: lti[lane * kLtiStride + field] -----------------------------
enum LaneI : int {
  LI_JTYPE = 0,   // 0 = none (base / padding lane), 1 revolute, 2 prismatic
  LI_PARENT = 1,  // parent lane, -1 for the base and padding lanes
  LI_LEVEL = 2,   // tree depth, -1 for padding lanes
  LI_JUMP = 3,    // kMaxRounds entries: ancestor at distance 2^k, -1 if beyond the base
  LI_CHILD = LI_JUMP + kMaxRounds,  // kMaxChildren entries: child lanes, -1 if none (child 0 = lane+1)
  LI_LINK = LI_CHILD + kMaxChildren,  // reference link index of this lane, -1 for padding lanes
  LI_JROW = LI_LINK + 1,              // joint row (link index - 1), -1 for the base / padding lanes
  LI_SUBTREE = LI_JROW + 1,           // links in the subtree of this lane's link (itself included): in the
                                      // depth-first lane order the subtree of lane j is lanes [j, j + size)
  LI_ANCHOR = LI_SUBTREE + 1,         // lane of the LEAF of this link's first-child chain: the reference point of its
                                      // articulated-body quantities (jxs_core.h, "Anchored ABA")
  LI_PANCHOR = LI_ANCHOR + 1,         // LI_ANCHOR of the parent link (own anchor for the base / first children)
  LI_RGPT = LI_PANCHOR + 1,           // rigid modes, merged Delassus sweeps (KParams::rg_merge): slot of the collidable
                                      // point whose parent link lies in the subtree of this link, -1 if none
  LI_COUNT = LI_RGPT + 1
};
constexpr int kLtiStride = (LI_COUNT + 3) / 4 * 4;  // staging table of the packer: one int per field
// [round 3] What the kernels read is PACKED: every field is a signed byte (lanes, levels, link and joint indices
// are all < 64 by construction), one 32-byte record per lane = two 128-bit loads instead of six.  The prologue of
// the step kernel is bound by the CU's vector-memory pipeline, not by latency (tools/ubench/cu_share.hip: a
// dwordx4 load costs the CU 16 ticks, a dword load 32, and the waves of a CU share the pipeline), so bytes fetched
// per lane are step time; unpacking is one v_bfe_i32 per USED field.
constexpr int kLtiPackWords = 8;  // dwords per lane in the packed table (LI_COUNT = 24 bytes used)
static_assert(LI_COUNT <= 4 * kLtiPackWords, "packed lane-int record too small");
JXS_HD constexpr int lti_word(int field) { return field >> 2; }  // word of the lane's record that holds the field
JXS_HD int lti_unpack(unsigned w, int field) { return (int)(w << (24 - 8 * (field & 3))) >> 24; }
JXS_HD int lti_get(const int* rec, int field) {  // rec: the lane's packed record
  return lti_unpack((unsigned)rec[lti_word(field)], field);
}

// ---- per-point-slot tables (slots = n_chunks * G), slot-major with stride 4: ptf[slot * 4 + field] ----
enum PointF : int { PF_POS = 0, PF_COUNT = 3 };
enum PointI : int {
  PI_BODY = 0,   // lane of the parent link, -1 for an empty slot
  PI_ROW = 1,    // original collidable-point index (row of the tangential deformation state)
  PI_TAIL = 2,   // number of slots after this one in the same (chunk, link) segment
  PI_L1 = 3,     // rigid modes (KParams::rg_merge): lane of the level-1 ancestor of the parent link
  PI_COUNT = 4
};
constexpr int kPtStride = 4;
// The point tables are stored per CHUNK of G slots, one fixed-size record per chunk (so that the address of
// the chunk-0 tables needs nothing but the table base and the compile-time G):
//   pti[G][kPtStride] int | ptf[G][kPtStride] T | head[G] int
// head[lane]: slot-lane of the first point of link `lane` in that chunk, -1 if none.
template <typename T>
JXS_HD constexpr int chunk_bytes(int G) { return G * kPtStride * 4 + G * kPtStride * (int)sizeof(T) + G * 4; }

// ---- row-distributed ABA: per-lane int table rti[lane * kRtiStride + field] ---
// In this phase lane = 8 * slot + row: the 8 lanes of a slot hold the 6 rows (2 idle) of the
// articulated inertia of ONE link per tree level; a first child inherits its parent's slot, so a
// serial chain never leaves its lanes.
constexpr int kRowLevels = 8;     // tree levels 0..7 (deeper trees use the link-per-lane sweeps)
constexpr int kRowExtra = 3;      // extra (non-first) children per link handled by cross-slot pulls
constexpr int kRowRec = 64;       // LDS words per link record (multiple of 16 bytes: 128-bit accesses)
enum RowI : int {
  RT_REC = 0,                         // [kRowLevels] LDS word offset of the record this row lane reads at level L:
                                      // link(L, slot)'s for its six row lanes, the all-zero record (lds_zero_rec) otherwise
  RT_FC = RT_REC + kRowLevels,        // bit L: link(L, slot) is the first child of link(L-1, slot)
  RT_PULL = RT_FC + 1,                // [kRowLevels][kRowExtra] lane to pull an extra child (level L) from, -1
  RT_PPULL = RT_PULL + kRowLevels * kRowExtra,  // [kRowLevels] lane holding the parent's row, -1 = same lane
  RT_COUNT = RT_PPULL + kRowLevels
};
constexpr int kRtiStride = (RT_COUNT + 3) / 4 * 4;  // staging table of the packer: one int per field
// packed form read by the kernels, one 64-byte record per lane (four 128-bit loads instead of eleven):
//   bytes 0..15  RT_REC[8] as unsigned 16-bit LDS word offsets | byte 16 RT_FC | bytes 17..40 RT_PULL[8][3] (signed
//   bytes) | bytes 41..48 RT_PPULL[8] (signed bytes)
constexpr int kRtiPackWords = 16;
JXS_HD constexpr int rti_byte(int field) { return field == RT_FC ? 16 : (field < RT_PPULL ? 17 + (field - RT_PULL) : 41 + (field - RT_PPULL)); }
JXS_HD constexpr int rti_word(int field) { return field < RT_FC ? (field >> 1) : (rti_byte(field) >> 2); }
JXS_HD int rti_unpack(unsigned w, int field) {
  if (field < RT_FC) return (int)((field & 1) ? (w >> 16) : (w & 0xffffu));
  return (int)(w << (24 - 8 * (rti_byte(field) & 3))) >> 24;
}
JXS_HD int rti_get(const int* rec, int field) { return rti_unpack((unsigned)rec[rti_word(field)], field); }
// LDS record layout (words).  Every group a lane reads together is 16-byte aligned and contiguous, because
// one ds instruction costs a lone wave ~14 cycles whether it moves 4 or 16 bytes (tools/ubench/issue_rate.hip):
//   8 r + 0..5  row r of M (6x6),  8 r + 6  pA[r],  8 r + 7  S[r]      (r < 6: what row lane r reads, two b128)
//   48..53 S, 54..59 c                                                   (slot-uniform, three b128)
//   60 tau, 61..63 anchor of this link's chain minus the anchor of its parent's chain (zero for first children)
//   sdd (the result of pass 3) takes the place of tau, which pass 2 has read by then: the record is 64 words
// [round 6] ONE RECORD PER LINK (lanes >= nL publish nothing) OF 64 WORDS, then kRowZero words of zeros -- 12.4 KB per
// humanoid wave instead of 18.3 KB (one 68-word record per LANE, 48 words of base rows, a 68-word zero record).  The LDS
// of gfx950 is handed out in granules of 1280 bytes: 12.4 KB is ten granules, TWELVE waves per CU = three per SIMD, which
// the kernel's 150 registers allowed all along (measured residency: profiles/r06_residency.txt).
//   * the base rows (6 x 8 words {row of MA_0, pA_0[r], -}) ALIAS the row region of record 0 -- the base link's own, read
//     for the last time at level 0, right before the base solve writes them (program order of a one-wave workgroup);
//   * a row lane without a link at a level reads "the record" at lds_zero_rec = Z - 48, Z = nL * kRowRec: its row group
//     at +48 instead of +8 row (one select per level), S | c at +48, tau | dp at +60, c_r at +54 + row, its sdd store
//     (a zero: 1 / d = 0 there) at +60 -- everything inside [Z, Z + 16).
enum RowLds : int { RL_ROW = 0, RL_ROW_PA = 6, RL_ROW_S = 7, RL_S = 48, RL_C = 54, RL_TAU = 60, RL_DP = 61, RL_SDD = RL_TAU };
constexpr int kRowZero = 16;  // zeros behind the records: what the lanes without a link read (and store: zeros)
// The link kinematics staged for the contact phase ([G][kKinRec] words) ALIAS the record area: the contact phase
// has read them back before the ABA publishes its records (a single-wave workgroup executes its LDS
// operations in program order).  20.6 KB -> 16 KB per humanoid wave: ten waves per CU instead of seven.
constexpr int kKinRec = 24;  // R (9), r (3), v_lin (3), v_ang (3), anchor of the link's chain (3), padding to whole 128-bit groups
JXS_HD constexpr int lds_kin_offset(int) { return 0; }
// row lanes without a link at a level read zeros instead of selecting them (nine v_cndmask per level saved): the
// "record" they read starts 48 words before the zero area (see above)
// Where record k starts.  A stride of 64 words would put the same word of every record into the same bank (the 24 link
// lanes of the humanoid publish with ONE address pattern: measured +40 % at 64 Ki environments, profiles/r06_experiments.md);
// the 68 of round 5 does not fit ten granules.  So: 64 words and a skew of four words (one 128-bit group) per PAIR of
// records -- 13 of the 16 alignments in use for 24 records, 48 words of padding per humanoid environment.
JXS_HD constexpr int lds_rec_off(int k) { return k * kRowRec + 4 * ((k + 1) >> 1); }
JXS_HD constexpr int lds_zero_at(int nL) { return lds_rec_off(nL); }
JXS_HD constexpr int lds_zero_rec(int nL) { return lds_zero_at(nL) - RL_S; }
JXS_HD constexpr int lds_base_rows(int) { return 0; }  // the base rows: the row region of record 0
// words per environment of the row layout of a model with nL links in groups of G lanes: the records and the zeros, or
// the kinematics staged for the contact phase ([G][kKinRec], aliasing the records), whichever is larger
JXS_HD constexpr int lds_rows_words(int G, int nL) {
  const int a = lds_zero_at(nL) + kRowZero, b = G * kKinRec, c = 64;
  return ((a > b ? (a > c ? a : c) : (b > c ? b : c)) + 3) / 4 * 4;
}
// upper bound over the models of a lane-group size (nL = G): where the offset of an area behind the row layout has to be
// a compile-time constant of G (two-wave workgroups, the RungeKutta4 chunk area)
JXS_HD constexpr int lds_words_per_env(int G) { return lds_rows_words(G, G); }
// [round 4] RungeKutta4 with more collidable points than lanes (SoftContacts): the points of the chunks behind the first
// go through memory at every stage (jxs_core.h contact_chunk); what RungeKutta4 carries between its stages for them --
// the deformation rate of the previous stage (3 words) and the weighted sum of the rates (3) -- sits in the LDS
// behind the area of the row layout, eight words per slot (six used: 128-bit LDS accesses want 16-byte alignment),
// written and read by the slot's own lane only
constexpr int kRk4SlotWords = 8;
JXS_HD constexpr int rk4_chunk_off(int G) { return lds_words_per_env(G); }
JXS_HD constexpr int rk4_lds_words_per_env(int G, int n_chunks) {
  return lds_words_per_env(G) + (n_chunks > 1 ? (n_chunks - 1) * G * kRk4SlotWords : 0);
}
// ---- two-wave workgroups: the inertia wave and the main wave work on the same
// environments; per environment the LDS holds
//   [0, W)        the inertia wave's records, base rows and zero record (the single-wave layout above)
//   [W, 2 W)      the main wave's records (same layout: its bias rows, S, c, tau), contact staging, base bias, zero record
//   [2 W, ...)    XL[level][lane][8]: what the inertia wave hands over per tree level -- row r of Ma = MA - U U^T / d
//                 (6 words), U_r, 1 / d -- read back by the same lane of the main wave
// and, behind the environments of the workgroup, kDuoFlagWords words of flags (levels published so far).
enum Role : int { ROLE_SOLO = 0, ROLE_MAIN = 1, ROLE_INERTIA = 2 };
//   [.., ...)     FK[lane][kDuoFkRec]: the forward kinematics of the inertia wave -- R (9), r (3), the anchor of the
//                 link's chain (3), anchor minus the parent chain's anchor (3) -- which the main wave does not repeat
//   base area of the inertia wave (48 words behind its records): the LDL^T factor of the articulated base inertia
constexpr int kDuoXlRec = 8;
constexpr int kDuoFkRec = 20;
constexpr int kDuoFlagWords = 4;
constexpr int kDuoFlagFk = 1;  // progress word: 1 = kinematics published, 1 + (kRowLevels - L) = level L published
JXS_HD constexpr int duo_main_off(int G) { return lds_words_per_env(G); }
JXS_HD constexpr int duo_xl_off(int G) { return 2 * lds_words_per_env(G); }
JXS_HD constexpr int duo_fk_off(int G) { return duo_xl_off(G) + kRowLevels * G * kDuoXlRec; }
JXS_HD constexpr int duo_words_per_env(int G) { return duo_fk_off(G) + G * kDuoFkRec; }
// rigid modes: Q and H packed lower triangles of order 3 n_cp + one exchange vector (jxs_rigid.inc)
// RelaxedRigidContacts (rigid == 2) factorises in place of the Delassus matrix: one triangle
// Problems of <= 4 points (the row-distributed register solver and the merged Delassus sweeps, jxs_rigid.inc) get
// 16 words for the solver's gather buffer (the H triangle may be shorter than it) and kRgMergeRec words per point
// for the exchange of the merged sweeps.  Larger problems get nothing extra: at 16 points four environments of a
// wave use 38.5 KB and four waves share a CU -- 176 more words per environment made it three (measured: the
// standing 16-point quadruped 0.83 -> 1.43 ms).
constexpr int kRgMergeRec = 40;  // [0, 18) wrench handed to the base per unit force, [20, 38) base acceleration
JXS_HD constexpr int rigid_lds_merge_off(int n_cp, int rigid = 1) {
  return ((rigid == 2 ? 1 : 2) * ((3 * n_cp * (3 * n_cp + 1)) / 2) + 3 * n_cp + 8 + (n_cp <= 4 ? 16 : 0) + 3) / 4 * 4;
}
// [round 5] RelaxedRigidContacts with more points than lanes (jxs_rigid.inc relaxed_contact_forces_chunked): one record of
// kCtRec words per point slot, read and written by the slot's own lane only
constexpr int kCtRec = 24;
JXS_HD constexpr int rigid_lds_words_per_env(int n_cp, int rigid = 1, int ct_tree = 0, int n_chunks = 1, int G = 0) {
  return ct_tree ? (n_chunks > 1 ? n_chunks * G * kCtRec : 16)  // [round 5] solved in the tree (jxs_rigid.inc ta_*): no triangle in the LDS
                 : rigid_lds_merge_off(n_cp, rigid) + (n_cp <= 4 ? 4 * kRgMergeRec : 0);
}
constexpr int kQpMaxIter = 30;    // interior-point iterations (oracle/refrigid.py QP_MAX_ITER)
constexpr int kRigidMaxPoints = 64;  // one lane per point: a full wave ([round 3]: 32 -> 64; the point masks are 64 bits wide)
constexpr int kDbgSlots = 64;      // developer profiling build: cycle stamps / counters per workgroup (the second wave of a two-wave workgroup stamps at +32)
constexpr int kImpactCgIters = 5;  // preconditioned CG iterations of the impact solve (jxs_rigid.inc)

enum Mode : int {
  MODE_STEP = 0,  // js.model.step                         api/model.py:2601-2681
  MODE_FD = 1,    // forward_dynamics_aba (no contacts)    api/model.py:1269-1406
  MODE_ID = 2,    // inverse_dynamics / RNEA               api/model.py:1746-1894
  MODE_KIN = 3,   // cached kinematics of JaxSimModelData  api/data.py:405-523
  MODE_ROLLOUT = 4,  // MODE_STEP repeated KArgs::n_steps times in one launch (state in registers)
  MODE_STEP_RK4 = 5,  // js.model.step with IntegratorType.RungeKutta4  api/integrators.py:91-167
  MODE_STEP_RIGID = 6,  // js.model.step with the RigidContacts / RelaxedRigidContacts model  rbda/contacts/rigid.py:176-539
  MODE_STEP_RK4_RIGID = 7,  // RungeKutta4 with RigidContacts / RelaxedRigidContacts (contact forces solved at every stage)
  MODE_CRBA = 8,  // free_floating_mass_matrix: composite-rigid-body algorithm  rbda/crba.py:10-170, api/model.py:1553-1590
  MODE_JAC = 9,   // doubly-left full Jacobian and its derivative  rbda/jacobian.py:128-339
  MODE_MINV = 10,  // free_floating_mass_matrix_inverse  rbda/mass_inverse.py:11-233, api/model.py:1593-1631
  MODE_GRAV = 11,  // joint torques of free_floating_gravity_forces (RNEA at zero velocity and acceleration,
                   // api/model.py:1897-1931): a dedicated kernel -- kinematics, subtree sums of the link weights, S . f
  // [round 6] js.ode.system_dynamics / system_acceleration (api/ode.py:16-131,174-225) and js.contact.link_contact_forces
  // (api/contact.py:514-603): the step up to -- not including -- the actuation model and the integrator.  Writes the
  // state DERIVATIVE in the layout of the state block (KArgs::state_out: pdot_B, Qdot, sdot, W_vdot_WB, sddot, mdot) and,
  // on request, the inertial 6D contact wrench of every link (KArgs::out_H, [nL * 6][N]).
  MODE_DYN = 12,        // SoftContacts / no collidable points
  MODE_DYN_RIGID = 13   // RigidContacts / RelaxedRigidContacts (contact forces solved like stage 0 of the step, no impact)
};
constexpr int kNumModes = 14;

enum ForceRepr : int { REPR_INERTIAL = 0, REPR_BODY = 1, REPR_MIXED = 2 };  // api/common.py:39-47

// Wave-uniform parameters, passed by value as the kernel argument (lives in SGPRs / kernarg).
template <typename T>
struct KParams {
  // topology
  int nL, n, n_points, n_slots, n_chunks, seg_steps, n_rounds, max_depth, floating, any_suc;
  // maxch(L): max #children (at level L) of any link at level L-1, 4 bits per level
  unsigned long long maxch_nib[(kMaxDepth + 1) / 16];
  unsigned long long nonadj_levels;  // bit L: some link at level L has its parent in a lane != lane-1
  int max_children;                  // [round 6] the widest link of the model (max over the nibbles above): a model-specialised kernel
                                     // keeps max_children child lanes in registers, not kMaxChildren (six more registers cost the rigid
                                     // contact kernels, which spill already, 6 - 23 %: profiles/r06_experiments.md)
  int seg_dpp_ok;                    // every (chunk, link) point segment lies inside one 16-lane row
  int row_mode;                      // 1: ABA passes run row-distributed (tables in KArgs::rti)
  unsigned int row_cross_levels;     // bit L: some parent pulls an extra child of level L across slots
  unsigned int row_ppull_levels;     // bit L: some link of level L has its parent in another slot
  unsigned int row_pull_counts;      // 4 bits per level L: max number of extra children (level L) any parent pulls
  int has_base_off;                  // 0: base_off = 0 (every floating-base URDF model): the terms of the base-link offset are left out
  int any_pri;                       // 1: the model has a prismatic joint (else their translation and motion-subspace terms are left out)
  // [round 3] link-per-lane sweeps: nibble k - 1, k = 1..5: EVERY link with a k-th child (a non-first child) finds it the
  // same number of lanes up, inside its own 16-lane row -- the child is gathered by a DPP row shift folded into the
  // accumulation instead of a ds_bpermute per value; 0: shuffle.  Used by the model-specialised kernels only (the shift
  // is an instruction modifier: a run-time value would be a fifteen-way branch around every gather).
  unsigned int child_off;
  JXS_HD int child_shift(int k) const { return (k >= 1 && k <= 5) ? (int)((child_off >> ((k - 1) * 4)) & 15u) : 0; }
  unsigned int row_pull_dpp;         // bit L * kRowExtra + k: every k-th extra child of level L sits in the slot next to its parent's, eight
                                     // lanes up in the same 16-lane row -- a DPP row shift reaches it (no ds_bpermute)  [round 3]
  JXS_HD int maxch(int L) const {
    // Constant indices and mask arithmetic only: a `L < 16 ? maxch_nib[0] : ...` chain is turned into a load
    // with a runtime index by the compiler, which puts this whole by-value struct into scratch memory and
    // every field read behind a scratch load (seen in the rigid-contact kernels: 296 bytes of scratch, loops
    // over P.n_cp / P.max_depth exec-masked because their bounds came back in VGPRs).
    const unsigned long long m0 = 0ull - (unsigned long long)(L < 16);
    const unsigned long long m1 = 0ull - (unsigned long long)(L >= 16 && L < 32);
    const unsigned long long m2 = 0ull - (unsigned long long)(L >= 32 && L < 48);
    const unsigned long long m3 = 0ull - (unsigned long long)(L >= 48);
    const unsigned long long w = (maxch_nib[0] & m0) | (maxch_nib[1] & m1) | (maxch_nib[2] & m2) | (maxch_nib[3] & m3);
    return (int)((w >> ((L & 15) * 4)) & 15ull);
  }
  // state-block rows ([row][N], N fastest): SURVEY.md section 8(a) row D
  int row_pos, row_quat, row_s, row_vlin, row_vang, row_sd, row_m, n_rows;
  // model constants
  T dt, g;                       // time_step, signed z gravity          api/model.py:54-60
  T K, D, mu, p, q, K_over_D;    // SoftContactsParams                   rbda/contacts/soft.py:24-46
  int pq_half;                   // p == q == 0.5 -> sqrt instead of pow
  T terrain_h;                   // terrain height over the origin       terrain/terrain.py:65-238
  T nrm[3];                      // PlaneTerrain unit normal (0,0,1 for FlatTerrain)
  int flat;                      // normal == +z: specialised contact arithmetic
  // [round 6] height-field terrain (include/jaxsim_amd.h jxs_model_desc::terrain_grid; reference: the generic Terrain of
  // terrain/terrain.py:15-62): heights hf[ix * hf_ny + iy] (KArgs::hf, behind the point chunks in the model block)
  int hf;                        // 1: the terrain is a height field (flat = 0 then)
  int hf_nx, hf_ny, hf_off;      // samples per axis; byte offset of the samples in the model block
  T hf_x0, hf_y0, hf_idx, hf_idy;  // origin, 1 / spacing
  T hf_delta, hf_inv_2delta;     // central-difference step of the normal (Terrain.delta = 0.010) and 1 / (2 delta)
  T tau_max, w_th, w_max;        // ActuationParams                      rbda/actuation/common.py:16-19
  T inv_w_range;                 // 1 / (w_max - w_th)
  int enable_friction;
  T base_off[3];                 // translation of suc_H_i[0] (quirk 12)
  T eps;                         // finfo(dtype).eps                     rbda/contacts/soft.py:246
  T quat_K;                      // Baumgarte gain of Quaternion.derivative (0.1)  math/quaternion.py:72
  // RigidContacts (rbda/contacts/rigid.py:95-174); K, D, mu above are then RigidContactsParams
  int rigid;                     // contact model: 0 SoftContacts, 1 RigidContacts, 2 RelaxedRigidContacts
  int n_cp;                      // enabled collidable points (= used slots of chunk 0 in the rigid modes)
  int rg_merge;                  // every point sits alone in its own subtree of the base: merged Delassus sweeps (jxs_rigid.inc)
  T reg_delassus;                // regularization_delassus (1e-6)
  T qp_tol;                      // solver_options["solver_tol"] (1e-3)
  T impact_rel_tol;              // relative Tikhonov shift of the impact preconditioner
  // RelaxedRigidContacts (rbda/contacts/relaxed_rigid.py:29-75, 540-591); K, D above then hold the
  // stiffness 1 / (d_max Omega zeta)^2 and damping 2 / (d_max Omega) of the reference acceleration
  T rr_dmin, rr_dmax, rr_inv_width, rr_mid, rr_pow;
  T rr_ca, rr_cb;                // 1 / mid^(p-1), 1 / (1-mid)^(p-1)
  T rr_rcoef;                    // 2 mu^2 (1 + mu^2)
  T rr_tiny;                     // smallest positive normal number (guards pow of a non-positive base)
  int rr_refine;                 // refinement steps against the operator applied through the tree
  // [round 5] contact problems solved IN THE TREE (jxs_rigid.inc ta_*): (J M^-1 J^T + D) x = c with D block diagonal per
  // point is x = D^-1 (c - J a), (M + J^T D^-1 J) a = J^T D^-1 c -- a forward-dynamics solve of the same tree whose contact
  // links carry the extra 6 x 6 "inertia" W_l = sum_p P_p^T D_p^-1 P_p.  No matrix, no rank decision, any number of contact
  // links.  1: this model's contact solves run that way (decided by the packer).
  int ct_tree;
  // [round 5] RigidContacts: the interior-point iteration starts from the UNCONSTRAINED minimiser x = -Q^-1 q instead of
  // CVXGEN's penalised one, and an environment in which that x is feasible is done without an iteration (jxs_rigid.inc).
  // 1 for models whose points sit alone in their own subtrees of a floating base (a legged robot with one point per foot:
  // config 5), where the Delassus matrix is well conditioned; with several points on one rigid body it is singular up to
  // the 1e-6 regularisation and its unconstrained minimiser is far from anything.
  int qp_warm;
  int jump_pad;                  // 1: pointer-jumping sources beyond the base point at a padding lane that holds the identity
                                 // transform and zero vectors (no selects in the rounds); 0: they are -1 (no padding lane: nL == G)
  // [round 3] The joint rows (and, with one point chunk of <= G collidable points, the deformation rows) of the state are
  // LOADED by lane index -- lane l >= 1 joint row l - 1, lane l point row l -- and permuted to the owning lanes when the
  // index tables have arrived (jxs_core.h, stage A).  1: that permutation is the identity, no shuffle.
  int jrow_seq, prow_seq;
  int rk4fast;                   // RungeKutta4Fast: contact forces and position derivatives of the initial state (api/integrators.py:170-276)
  int anchored;                  // 1: the ABA of step / forward dynamics refers every first-child chain to its leaf link (fp32 conditioning)
};

// Device/host pointers handed to the core for one launch.
enum Knob : int { KNOB_NO_MFMA = 1, KNOB_DUO = 2, KNOB_NO_COMMON_VARIANT = 4 };
template <typename T>
struct KArgs {
  const T* ltf;        // [G][kLtfStride]
  const int* lti;      // [G][kLtiPackWords] packed lane-int records (lti_get)
  const unsigned char* chunks;  // [max(n_chunks, 1)] point-chunk records (chunk_bytes<T>(G) each)
  const int* rti;      // [G][kRtiPackWords] packed row-distributed ABA tables (rti_get; row_mode only)
  const T* state_in;   // [n_rows][N]
  T* state_out;        // [n_rows][N] (may alias state_in)
  const T* tau;        // [n][N] or null            joint_force_references / joint_forces
  const T* link_f;     // [nL*6][N] or null         link_forces
  int force_repr;      // ForceRepr of link_f
  const T* in_a;       // MODE_ID: [6+n][N] inertial base acceleration + joint accelerations, or null
  T* out_a;            // MODE_FD: [6+n][N] ; MODE_ID: [6+n][N] (base wrench, joint torques)
  T* out_H;            // MODE_KIN: [nL*12][N] rows of [R|p] per link (row-major 3x4); MODE_DYN*: [nL*6][N] inertial link contact wrenches, or null
  T* out_V;            // MODE_KIN: [nL*6][N] inertial-fixed link velocities
  int N;               // batch size (leading dimension of every [row][N] array)
  int n_steps;         // MODE_STEP: consecutive steps fused in this launch (state carried in registers)
  T* out_tau;          // MODE_ID, optional: joint torques only, [n][N] (the layout `tau` is read in)
  int id_zero_vel;     // MODE_ID: evaluate at zero velocity (gravity term g(q), api/model.py:1897-1931)
  long long* dbg;      // developer builds (-DJXS_PHASE_TIMING): [blocks][kDbgSlots] cycle stamps, else null
  int* faults;         // [2] environments whose QP contact-force solve / impact solve was discarded (non-finite), or null
  int flags;           // switches of a launch: bit 0 = no MFMA in the contact solvers' Cholesky (developer A/B against the vector
                       // path); bit 1 = gravity-compensated step of the rigid contact modes (tau_ref += g(q))
                       // bit 2 = MODE_ROLLOUT: `tau` is a sequence, one [n][N] block of rows per fused step (jxs_rollout_controlled)
  int spec_consts;     // 1: the integer model flags of KParams are compile-time constants in this kernel (jxs_spec.hip)
  int knobs;           // host only: developer knobs of the launcher (KNOB_*), read from the environment ONCE by the
                       // library (jxs_api.hip debug_knobs; jxs_debug_reload_env re-reads them for the tests)
  int duo_max_blocks;  // host only: largest grid the two-wave variant is used for (JXS_DUO_MAX_BLOCKS)
  const T* hf;         // height-field samples [hf_nx][hf_ny] (model block + KParams::hf_off), or null
  T fparam;            // MODE_DYN / MODE_DYN_RIGID: Baumgarte gain of the quaternion derivative (api/ode.py:136-169; default 1.0)
  int has_lds;         // the launch has the per-environment LDS area of the row layout (known when the wave starts: a
                       // compile-time constant in the specialised / common-feature kernels): the thirteen
                       // environment-uniform rows of the state are fetched by ONE load instruction and spread through it
};

// ---- device model block: ONE allocation per model that the kernels address from a single pointer -------
//   KParams<T> (padded to 256 B) | ltf | lti | rti | point-chunk records
// The wave-uniform parameters are read from it with scalar loads (device memory, L2-resident after the
// first wave); round 1 passed them by value as kernel arguments, and the kernarg segment turned out to be a
// ~1600-cycle round trip per wave (measured with arrival stamps).
template <typename T>
JXS_HD constexpr int mblk_off_ltf() { return ((int)sizeof(KParams<T>) + 255) / 256 * 256; }
template <typename T>
JXS_HD constexpr int mblk_off_lti(int G) { return mblk_off_ltf<T>() + G * kLtfStride * (int)sizeof(T); }
template <typename T>
JXS_HD constexpr int mblk_off_rti(int G) { return mblk_off_lti<T>(G) + G * kLtiPackWords * 4; }
template <typename T>
JXS_HD constexpr int mblk_off_chunks(int G) { return mblk_off_rti<T>(G) + G * kRtiPackWords * 4; }

}  // namespace jxs
