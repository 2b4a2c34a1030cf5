/* jaxsim_amd -- C ABI of the MI355X-native batched rigid-body step.
 *
 * Drop-in boundary for the hot path of ami-iit/jaxsim.  The reference has no FFI seam: the
 * path sits behind plain Python functions taking (model, data) pytrees.  Each entry point
 * below names the reference function it replaces (paths relative to the reference root).
 * A Python maintainer binds these with ctypes (INTEGRATION.md shows the stub); the in-tree
 * binding is jaxsim_amd/_lib.py.
 *
 * Conventions
 *  - plain C types only; no torch / HIP types in signatures (streams are `void*`
 *    = hipStream_t, NULL = default stream);
 *  - every batched array is a *device* pointer, struct-of-arrays with the batch index fastest,
 *    tile-interleaved: written below as [rows][N], stored as [ceil(N/T)][rows][T] where
 *    T = jxs_layout.tile environments (the environments one wavefront processes; T = 2 for the
 *    24-link humanoid).  Element (row, env) lives at ((env/T)*rows + row)*T + env%T; arrays are
 *    allocated in whole tiles.  dtype = the model's dtype (float or double);
 *  - state block rows (reference `JaxSimModelData`, src/jaxsim/api/data.py:46-63; the base
 *    velocity is stored inertial-fixed like the reference, :151-156,187-188):
 *        base_position[3] base_quaternion[4] (wxyz) joint_positions[n]
 *        base_linear_velocity[3] base_angular_velocity[3] joint_velocities[n]
 *        tangential_deformation[n_cp][3]
 *    row offsets are reported by jxs_model_layout();
 *  - all kernels are asynchronous on the given stream; the caller synchronises;
 *  - functions return 0 on success, a negative JXS_E* code otherwise, and
 *    jxs_last_error() returns a thread-local message.  No exceptions cross the ABI.
 */
#ifndef JAXSIM_AMD_H
#define JAXSIM_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JXS_OK 0
#define JXS_EINVAL (-1)      /* bad argument / unsupported model                     */
#define JXS_ENODEVICE (-2)   /* no HIP device / HIP runtime error                    */
#define JXS_ENOMEM (-3)
#define JXS_ECOMM (-4)       /* RCCL error                                           */

#define JXS_F32 0
#define JXS_F64 1

/* Velocity / force representations: src/jaxsim/api/common.py:39-47 */
#define JXS_REPR_INERTIAL 0
#define JXS_REPR_BODY 1
#define JXS_REPR_MIXED 2

/* IntegratorType: src/jaxsim/api/model.py:32-40.  RungeKutta4Fast (api/integrators.py:170-276) only with
 * the rigid contact models: the reference's version corrupts the SoftContacts state and fails without
 * collidable points. */
#define JXS_INTEGRATOR_SEMI_IMPLICIT_EULER 0
#define JXS_INTEGRATOR_RUNGE_KUTTA4 1
#define JXS_INTEGRATOR_RUNGE_KUTTA4_FAST 2

/* Contact models: src/jaxsim/rbda/contacts/{soft,rigid,relaxed_rigid}.py */
#define JXS_CONTACT_SOFT 0
#define JXS_CONTACT_RIGID 1
#define JXS_CONTACT_RELAXED_RIGID 2

/* Host description of one model: the static tables of `KinDynParameters`
 * (src/jaxsim/api/kin_dyn_parameters.py:86-284) plus the model-level constants of
 * `JaxSimModel` (src/jaxsim/api/model.py:52-82).  All arrays are host pointers, indexed by
 * the reference link index (entry 0 of the per-joint arrays is unused: joint i moves link i).
 */
typedef struct jxs_model_desc {
  int32_t n_links;
  int32_t floating_base;
  int32_t dtype; /* JXS_F32 | JXS_F64 */
  const int32_t* parent;        /* [nL]   parent_array, parent[0] = -1                      */
  const int32_t* joint_type;    /* [nL]   1 revolute, 2 prismatic (math/joint_model.py)     */
  const double* joint_axis;     /* [nL][3]                                                  */
  const double* lambda_H_pre;   /* [nL][16] row-major 4x4                                   */
  const double* suc_H_i;        /* [nL][16]                                                 */
  const double* link_mass;      /* [nL]                                                     */
  const double* link_com;       /* [nL][3]                                                  */
  const double* link_inertia;   /* [nL][9] I_CoM row-major                                  */
  const double* friction_static;       /* [nL] */
  const double* friction_viscous;      /* [nL] */
  const double* position_limit_min;    /* [nL] */
  const double* position_limit_max;    /* [nL] */
  const double* position_limit_spring; /* [nL] */
  const double* position_limit_damper; /* [nL] */
  int32_t n_points;               /* collidable points (ContactParameters, :765-840)        */
  const int32_t* point_body;      /* [n_points]                                             */
  const double* point_position;   /* [n_points][3]                                          */
  const uint8_t* point_enabled;   /* [n_points]                                             */
  double time_step;               /* api/model.py:54-56                                     */
  double gravity;                 /* signed z acceleration (-9.81)                          */
  double K, D, mu, p, q;          /* SoftContactsParams, rbda/contacts/soft.py:24-46        */
  double terrain_height;          /* Flat/PlaneTerrain height over the origin, terrain/terrain.py:65-238 */
  double torque_max, omega_th, omega_max; /* ActuationParams, rbda/actuation/common.py:16-19 */
  int32_t enable_friction;
  double terrain_normal[3];       /* PlaneTerrain unit normal (terrain/terrain.py:127-238); (0,0,1) = FlatTerrain */
  int32_t integrator;             /* JXS_INTEGRATOR_*: model.integrator, api/model.py:2665-2678 */
  int32_t contact_model;          /* JXS_CONTACT_*; with JXS_CONTACT_RIGID K, D, mu are RigidContactsParams
                                     (rbda/contacts/rigid.py:28-42) and p, q are ignored */
  double regularization_delassus; /* RigidContacts.regularization_delassus (rigid.py:99-101), 1e-6 */
  double solver_tol;              /* RigidContacts solver_options["solver_tol"] (rigid.py:103-108), 1e-3 */
  /* JXS_CONTACT_RELAXED_RIGID: RelaxedRigidContactsParams (rbda/contacts/relaxed_rigid.py:29-75); mu above is
     its friction coefficient, K and D are ignored exactly as the reference ignores them (:567-568) */
  double rr_time_constant, rr_damping_coefficient, rr_d_min, rr_d_max, rr_width, rr_midpoint, rr_power;
  /* [round 6] Height-field terrain: the reference's generic `Terrain` (terrain/terrain.py:15-62: any height(x, y), the
     normal by central differences with delta = 0.01) in the form a C ABI can carry -- heights sampled on a regular
     grid, sample [ix][iy] at (origin + (ix dx, iy dy)), row-major with x outer; height(x, y) is the BILINEAR
     interpolant (clamped to the border samples outside the grid) and the normal is
     n = [(h(x-d,y) - h(x+d,y)) / 2d, (h(x,y-d) - h(x,y+d)) / 2d, 1] / |.| of THAT function, d = terrain_delta
     (terrain.py:52-62).  NULL: the flat / plane terrain above.  With a grid `terrain_height` / `terrain_normal` are
     not used.  All three contact models.                                                                           */
  const double* terrain_grid;     /* [terrain_nx][terrain_ny] or NULL                                               */
  int32_t terrain_nx, terrain_ny; /* >= 2 each                                                                      */
  double terrain_origin[2];       /* (x, y) of sample [0][0]                                                        */
  double terrain_spacing[2];      /* (dx, dy), both > 0                                                             */
  double terrain_delta;           /* step of the central difference (Terrain.delta = 0.010, terrain.py:23); 0 = 0.01 */
} jxs_model_desc;

typedef struct jxs_model jxs_model; /* opaque, immutable after creation, shareable */

typedef struct jxs_layout {
  int32_t n_links, n_joints, n_points, n_rows;
  int32_t row_pos, row_quat, row_s, row_vlin, row_vang, row_sd, row_m;
  int32_t group; /* lanes per environment chosen for this model */
  int32_t tile;  /* T = 64 / group: environments per tile of every batched array */
  int32_t dtype;
  int32_t row_mode; /* 1: ABA passes run row-distributed (8 lanes per active link), 0: link per lane */
} jxs_layout;

/* ---- library / device ---------------------------------------------------------------- */
const char* jxs_last_error(void);
int jxs_device_count(int* count);
int jxs_set_device(int device);
int jxs_malloc(void** dptr, uint64_t bytes);
int jxs_free(void* dptr);
int jxs_memcpy_h2d(void* dst, const void* src, uint64_t bytes, void* stream);
int jxs_memcpy_d2h(void* dst, const void* src, uint64_t bytes, void* stream);
int jxs_memcpy_d2d(void* dst, const void* src, uint64_t bytes, void* stream);
int jxs_memset(void* dst, int value, uint64_t bytes, void* stream);
int jxs_stream_create(void** stream);
int jxs_stream_destroy(void* stream);
int jxs_stream_synchronize(void* stream);
/* Same completion guarantee, waited for by polling from the calling thread (microsecond-scale wake-up;
 * for timing short regions).                                                                   */
int jxs_stream_wait_spin(void* stream);
int jxs_device_synchronize(void);
/* HIP events on a stream (bench.py times the step kernel with these). */
int jxs_event_create(void** event);
int jxs_event_destroy(void* event);
int jxs_event_record(void* event, void* stream);
int jxs_event_elapsed_ms(void* start, void* stop, float* ms);

/* ---- model ----------------------------------------------------------------------------
 * Replaces JaxSimModel.build / KinDynParameters.build as far as the device is concerned:
 * uploads the constant tables (src/jaxsim/api/model.py:225-330).                        */
int jxs_model_create(const jxs_model_desc* desc, jxs_model** out);
int jxs_model_destroy(jxs_model* model);
int jxs_model_layout(const jxs_model* model, jxs_layout* out);

/* ---- model-specialised kernels ----------------------------------------------------------
 * The reference compiles `step` per model: the kinematic tree (parent array, joint types, ...) is static
 * under jax.jit (src/jaxsim/api/model.py:36-120, `static_field`s of KinDynParameters) and XLA folds it into
 * the program.  The generic kernels of this library read the same facts as wave-uniform flags at run time
 * and branch on them, which costs a lone wave ~20 % of the step.  A specialised
 * kernel is the SAME hand-written kernel source compiled with those integer flags as constants
 * (jaxsim_amd/csrc/jxs_spec.hip, built by jaxsim_amd/specialize.py with hipcc); physical parameters
 * (masses, gains, time step, contact parameters) stay run-time data.
 *
 * jxs_kernel_spec: the canonical description of the flags a kernel of `mode` (JXS_MODE_* below) would be
 *   specialised on, as text "T=float;G=32;MODE=0;P.n_chunks=1,P.seg_steps=4,...".  Host-only (no device
 *   needed).  Returns the length written (excluding the terminator) or a negative error code.
 * jxs_model_attach_specialized: loads a shared object built for exactly that description (checked) and
 *   routes the launches of `mode` for this model through it.  Cached launch graphs are dropped.     */
enum { JXS_MODE_STEP = 0, JXS_MODE_STEP_RIGID = 6 };
int jxs_kernel_spec(const jxs_model_desc* desc, int mode, char* buf, int capacity);
int jxs_model_attach_specialized(jxs_model* model, int mode, const char* shared_object_path);
/* bit `mode` set: launches of that mode use a specialised kernel */
int jxs_model_specialized_modes(const jxs_model* model, unsigned* mask);

/* ---- hot path ------------------------------------------------------------------------- */

/* js.model.step (src/jaxsim/api/model.py:2601-2681): actuation model -> soft contacts ->
 * ABA -> semi-implicit Euler.  state_out may alias state_in (in-place).  `tau` =
 * joint_force_references [n][N] or NULL (zeros); `link_forces` = [nL*6][N] or NULL,
 * expressed in `force_repr` (the data's velocity representation, api/model.py:2641-2646). */
int jxs_step(jxs_model* model, const void* state_in, void* state_out, const void* tau,
             const void* link_forces, int force_repr, int N, void* stream);

/* `n_launches` back-to-back in-place jxs_step launches enqueued from one call (no fusion: one kernel
 * launch per step, exactly what a host loop over jxs_step enqueues, without the per-call cost of the
 * host language).  On a created stream blocks of 250 and of 50 launches are captured once into hipGraphs
 * and replayed while the arguments stay the same; a remainder below 50 is launched plainly (faster to get
 * going than a graph of that size: tools/region_overhead.py).                                   */
int jxs_step_repeat(jxs_model* model, void* state, const void* tau, const void* link_forces,
                    int force_repr, int N, int n_launches, void* stream);

/* jxs_step_repeat bracketed by stream synchronisations on both sides (polling waits) and a host wall clock,
 * all inside one call: the timed region of a benchmark without the interpreter's call overhead inside it
 * (bench.py times its regions of exactly --steps launches with this).  *seconds = wall time of the region. */
int jxs_step_repeat_timed(jxs_model* model, void* state, const void* tau, const void* link_forces,
                          int force_repr, int N, int n_launches, void* stream, double* seconds);

/* `n_steps` consecutive steps with constant inputs in ONE launch sequence (what a
 * `jax.lax.fori_loop` over `step` does in the reference's notebooks); in place.           */
int jxs_rollout(jxs_model* model, void* state, const void* tau, const void* link_forces,
                int force_repr, int N, int n_steps, void* stream);

/* [round 4] `n_steps` consecutive steps with a SEQUENCE of joint torques -- what a `jax.lax.scan` of `step` over
 * precomputed `joint_force_references` does (open-loop rollouts: trajectory sampling, MPPI); in place.
 * `tau_seq` = [n_steps * n][N] in the usual [row][N] convention: rows k*n .. (k+1)*n-1 are the torques of step k
 * (the actuation model is applied to them at every step, like `step`).  `link_forces` stay constant.  Where the
 * steps fuse (semi-implicit Euler, SoftContacts, one chunk of collidable points) this is ONE launch with the state in
 * registers and one torque load per step; otherwise one launch per step, the step's torques gathered by a strided
 * device copy into a scratch block the MODEL owns: unfused controlled rollouts of one model are not re-entrant
 * across streams (use one model handle per stream, as for every call that carries per-model device state).   */
int jxs_rollout_controlled(jxs_model* model, void* state, const void* tau_seq, const void* link_forces,
                           int force_repr, int N, int n_steps, void* stream);

/* [round 4] A rollout that RECORDS: the state block after every step goes to `out_states` = [n_steps * n_rows][N]
 * (rows k*n_rows .. (k+1)*n_rows-1 = the state after step k, the layout of `state`), what `jax.lax.scan` over `step`
 * returns as its stacked outputs; `state` is advanced in place as by jxs_rollout.  `tau`: constant [n][N] or NULL, or
 * -- `tau_per_step` != 0 -- a sequence [n_steps * n][N] as in jxs_rollout_controlled.  Fused into one launch where the
 * steps fuse (one store of the state per step from registers); otherwise one launch and one strided device copy
 * per step.                                                                                     */
int jxs_rollout_recorded(jxs_model* model, void* state, const void* tau, int tau_per_step, const void* link_forces,
                         int force_repr, int N, int n_steps, void* out_states, void* stream);

/* forward_dynamics_aba (src/jaxsim/api/model.py:1269-1406) in inertial representation:
 * out_acc = [6+n][N] = inertial-fixed base acceleration then joint accelerations.
 * `joint_forces` are applied as given (no actuation model), no contact forces.          */
int jxs_forward_dynamics_aba(jxs_model* model, const void* state, const void* joint_forces,
                             const void* link_forces, int force_repr, void* out_acc, int N,
                             void* stream);

/* [round 6] system_dynamics / system_acceleration (src/jaxsim/api/ode.py:16-131,174-225) and link_contact_forces
 * (src/jaxsim/api/contact.py:514-603) -- what the reference's contact-model benchmarks time
 * (tests/test_benchmark.py:103-139) -- in ONE launch: contact forces of the model's contact model (SoftContacts,
 * RigidContacts, RelaxedRigidContacts), summed per link, plus the external `link_forces`, through ABA.  No actuation
 * model (`joint_torques` are applied as given, api/ode.py:117-122), no integrator, no impact.
 *   out_xdot  = [n_rows][N], the layout of the state block, every row holding the time derivative of the state row
 *               it stands for, inertial-fixed like the state: pdot_B = v_W + w x p_B, Qdot (Quaternion.derivative of
 *               the normalised quaternion with the Baumgarte gain `baumgarte`, api/ode.py:136-169; the reference's
 *               default is 1.0), sdot, W_vdot_WB (linear, angular), sddot, and the rate of the tangential deformation
 *               of the enabled points (zero for disabled points and for the rigid contact models).  May be NULL
 *               when only the link wrenches are wanted (SoftContacts then stops before ABA).
 *   out_link_contact_forces = [nL * 6][N] or NULL: the 6D contact wrench of every link, inertial representation
 *               ([f; p_W x f] summed over the link's enabled points, api/contact.py:592-601), rows 6 l .. 6 l + 5.
 * `link_forces` / `force_repr` as in jxs_step.                                                              */
int jxs_system_dynamics(jxs_model* model, const void* state, const void* joint_torques, const void* link_forces,
                        int force_repr, double baumgarte, void* out_xdot, void* out_link_contact_forces, int N,
                        void* stream);
/* link_contact_forces alone (api/contact.py:514-555): jxs_system_dynamics with out_xdot = NULL; `out_mdot`
 * ([3 * n_points][N] or NULL) receives the `m_dot` entry of the reference's aux dictionary for SoftContacts (rows
 * 3 c .. 3 c + 2 = point c; zeros for disabled points and for the rigid contact models).                        */
int jxs_link_contact_forces(jxs_model* model, const void* state, const void* joint_torques, const void* link_forces,
                            int force_repr, void* out_link_contact_forces, void* out_mdot, int N, void* stream);

/* inverse_dynamics / RNEA (src/jaxsim/api/model.py:1746-1894, rbda/rnea.py:12-238) in
 * inertial representation: in_acc = [6+n][N] (base acceleration, joint accelerations) or
 * NULL (zeros => free_floating_bias_forces, :1934-1978); out = [6+n][N] = base wrench then
 * joint torques.                                                                        */
int jxs_inverse_dynamics(jxs_model* model, const void* state, const void* in_acc,
                         const void* link_forces, int force_repr, void* out_forces, int N,
                         void* stream);

/* Layout conversion on the device between an environment-major array `[N][rows]` (row-major, what the
 * reference's vmapped arrays, a NumPy upload or a DLPack / __cuda_array_interface__ buffer of another
 * framework hold) and the tile-interleaved storage every batched argument of this library uses
 * (`jxs_layout.tile`).  The tiled buffer must hold ceil(N / tile) * rows * tile elements.      */
int jxs_tile_from_env_major(const void* src, void* dst, int rows, int N, int tile, int dtype, void* stream);
int jxs_tile_to_env_major(const void* src, void* dst, int rows, int N, int tile, int dtype, void* stream);

/* Value checks of a state block, the counterpart of the host callbacks the reference enables with
 * JAXSIM_ENABLE_EXCEPTIONS (src/jaxsim/exceptions.py:6-60, rbda/utils.py:135-146).  Synchronous.
 * counts3[0] = environments whose base quaternion contains NaN, [1] = environments whose quaternion is
 * not normalised (jnp.allclose(q.q, 1)), [2] = environments with any non-finite state entry.      */
int jxs_validate_state(jxs_model* model, const void* state, int N, int* counts3, void* stream);

/* RigidContacts only.  A contact-force QP or an impact solve whose result is not finite (fp32 on a numerically
 * singular stance) is DISCARDED -- that environment takes the step without contact forces / without the
 * velocity reset instead of poisoning its state -- where the reference would propagate NaN
 * (src/jaxsim/rbda/contacts/rigid.py:331-379).  The discards are counted per model: counts2[0] = contact-force
 * solves, counts2[1] = impact solves, in environments, since the last reset.  Synchronous.          */
int jxs_solver_fault_counts(jxs_model* model, int* counts2, int reset, void* stream);

/* [round 4] jxs_step with tau_ref + g(q): the joint part of free_floating_gravity_forces (src/jaxsim/api/model.py:1897-1931)
 * of the INPUT state is added to the joint force references inside the step kernel -- the controller loop
 *     tau = js.model.free_floating_gravity_forces(model, data)[6:] (+ tau_user);  data = js.model.step(model, data, joint_force_references=tau)
 * (BASELINE config 5) in one launch instead of two.  `tau` may be null (pure gravity compensation) or hold tau_user.
 * Models with a rigid contact model (RigidContacts / RelaxedRigidContacts and enabled points) only: JXS_EINVAL otherwise.      */
int jxs_step_gravity_compensated(jxs_model* model, const void* state_in, void* state_out, const void* tau,
                                 const void* link_forces, int force_repr, int N, void* stream);

/* Developer knobs of the launcher (JXS_DUO, JXS_DUO_MAX_BLOCKS, JXS_NO_MFMA, JXS_DISABLE_COMMON_VARIANT: A/B switches
 * between kernel variants that compute the same thing) are read from the environment once per process, at the first
 * launch.  This call reads them again -- for test programs that switch variants between launches.  No reference
 * counterpart (XLA flags are read once too).                                                          */
int jxs_debug_reload_env(void);

/* Gravity compensation torques: the joint part of free_floating_gravity_forces
 * (src/jaxsim/api/model.py:1897-1931: RNEA at zero velocity, zero acceleration, no external forces),
 * written as [n][N] -- the layout jxs_step reads `tau` in, so a controller loop
 * "tau = g(q); step(tau)" (BASELINE.json config 5) stays on the device.                     */
int jxs_gravity_torques(jxs_model* model, const void* state, void* out_tau, int N, void* stream);

/* free_floating_mass_matrix (src/jaxsim/api/model.py:1553-1590, rbda/crba.py:10-170): the composite-rigid-body
 * algorithm, one launch.  out_M = [(6+n)*(6+n)][N], row-major (row r, column c at row index r*(6+n)+c), in
 * MIXED velocity representation (base block about the base position, world axes); the Body / Inertial forms
 * are the 6x6 block congruence of api/model.py:1529-1551 applied by the caller.                  */
int jxs_mass_matrix(jxs_model* model, const void* state, void* out_M, int N, void* stream);

/* free_floating_mass_matrix_inverse (src/jaxsim/api/model.py:1593-1631, rbda/mass_inverse.py:11-233): the columns
 * are the responses of the articulated-body factorisation to unit generalized forces, one launch.  Same
 * layout and representation (MIXED) as jxs_mass_matrix; the six base rows / columns of a fixed-base model are
 * zero (a fixed base does not accelerate).                                                       */
int jxs_mass_matrix_inverse(jxs_model* model, const void* state, void* out_Minv, int N, void* stream);

/* jacobian_full_doubly_left + jacobian_derivative_full_doubly_left (src/jaxsim/rbda/jacobian.py:128-339), one
 * launch: out_J = [2*6*(6+n)][N] = B_J_full_WX_B (6 x (6+n), row-major) followed by B_Jdot_full_WX_B, both with
 * input and output in the base frame ("doubly left"); out_B_H_L = [nL*12][N] rows of [R|p] of every link
 * relative to the base, or NULL.  The link Jacobians / their derivatives in any pair of representations are
 * the column masks and 6x6 transforms of api/model.py:925-1228 applied by the caller.             */
int jxs_jacobian_full(jxs_model* model, const void* state, void* out_J, void* out_B_H_L, int N, void* stream);

/* The cached kinematics of JaxSimModelData.replace (src/jaxsim/api/data.py:405-523,
 * rbda/forward_kinematics.py:12-113): link transforms [nL*12][N] (rows of [R|p]) and
 * inertial-fixed link velocities [nL*6][N]; either output may be NULL.                  */
int jxs_refresh_kinematics(jxs_model* model, const void* state, void* out_link_transforms,
                           void* out_link_velocities, int N, void* stream);

/* ---- multi-GPU: one process per GPU, batch sharded, no per-step communication ---------
 * One RCCL all-gather over xGMI concatenates the final state shards (SURVEY.md section
 * 8(e)); the reference has no counterpart (single-device JAX).  The 128-byte unique id is
 * created on rank 0 and distributed by the host launcher (torch.distributed / MPI / file). */
int jxs_comm_unique_id(char id[128]);
int jxs_comm_init(void** comm, const char id[128], int rank, int world_size);
int jxs_comm_destroy(void* comm);
/* [round 5] what a scaling record must be able to say about its communicator: the version of the RCCL library the
 * ranks talk through (ncclGetVersion: e.g. 22107) and the PCI bus id of the device a rank runs on ("0000:05:00.0";
 * buf of at least 16 bytes) -- bench.py puts both into its line and refuses a run whose communicator is not RCCL
 * when asked to (--require-rccl).  */
int jxs_comm_version(int* version);
int jxs_device_pci_bus_id(char* buf, int len);
/* recv[world][rows][n_local]  <-  send[rows][n_local] of every rank */
int jxs_allgather(void* comm, const void* send, void* recv, uint64_t count, int dtype,
                  void* stream);

#ifdef __cplusplus
}
#endif
#endif /* JAXSIM_AMD_H */
