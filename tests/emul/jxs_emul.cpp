// TEST INFRASTRUCTURE: CPU lockstep emulation of the HIP kernel core.
//
// Compiles jaxsim_amd/csrc/jxs_core.h -- the very source of the gfx950 kernels -- against the
// host lane backend (jxs_lanes_host.h) so that `pytest -m "not gpu"` can compare the kernel
// logic (table packing, shuffles along the tree, level loops, contacts, integrator) with the
// oracle on a machine without a GPU.  Not part of the product: jaxsim_amd/ never loads it.
#include <algorithm>
#include <vector>
#include <cstring>
#include <string>

#include "jxs_lanes_host.h"
// lanes first: the core's unqualified calls on Vec resolve by ADL
#include "../../jaxsim_amd/csrc/jxs_core.h"
#include "../../jaxsim_amd/csrc/jxs_pack.h"

// The emulation is built as several translation units in parallel (tests/emul_binding.py): one per (dtype, lanes per
// environment) -- compiled with -DJXS_EMUL_UNIT_T=<float|double> -DJXS_EMUL_UNIT_G=<4..64>, each holds the explicit
// instantiation of run_group for that pair (the kernel core in all its modes) -- and the main unit (no macro) with the
// packer, the dispatch and the C entry points.  One g++ run over everything took 3.5 minutes.
namespace jxs_emul {

#ifdef JXS_EMUL_UNIT_G
extern thread_local std::string g_err;
extern void* g_dbg_ptr;
#else
thread_local std::string g_err;
void* g_dbg_ptr = nullptr;
#endif

template <typename T, int G>
void run_group(const jxs::Packed<T>& pk, jxs::KArgs<T> a, int mode);

#ifdef JXS_EMUL_UNIT_G
template <typename T, int G>
void run_group(const jxs::Packed<T>& pk, jxs::KArgs<T> a, int mode) {
  a.ltf = pk.ltf.data();
  a.lti = pk.lti_packed.data();
  a.chunks = pk.chunks.data();
  a.rti = pk.rti_packed.data();
  a.hf = pk.hf.empty() ? nullptr : pk.hf.data();
  for (int env = 0; env < a.N; ++env) {
    // what the device launch of this mode allocates per environment (jxs_kernels.h launch_one)
    const bool rigid_mode = mode == jxs::MODE_STEP_RIGID || mode == jxs::MODE_STEP_RK4_RIGID || mode == jxs::MODE_DYN_RIGID;
    const bool rows_mode = pk.P.row_mode && (mode == jxs::MODE_STEP || mode == jxs::MODE_ROLLOUT || mode == jxs::MODE_FD || mode == jxs::MODE_DYN);
    size_t limit = (size_t)-1;  // (modes without an LDS area of their own use it only where has_lds says so: not checked)
    if (rigid_mode) limit = (size_t)jxs::rigid_lds_words_per_env(pk.P.n_cp, pk.P.rigid, pk.P.ct_tree, pk.P.n_chunks, G);
    else if (mode == (jxs::MODE_STEP | 0x100)) limit = (size_t)jxs::duo_words_per_env(G);
    else if (mode == jxs::MODE_STEP_RK4 && (pk.P.row_mode || pk.P.n_chunks > 1)) limit = (size_t)jxs::rk4_lds_words_per_env(G, pk.P.n_chunks);
    else if (rows_mode) limit = (size_t)jxs::lds_rows_words(G, pk.P.nL);
    jxs::HostLanes<T, G> ln(a.N, env, (size_t)std::max({jxs::rigid_lds_words_per_env(pk.P.n_cp, pk.P.rigid), jxs::rigid_lds_words_per_env(pk.P.n_cp, pk.P.rigid, pk.P.ct_tree, pk.P.n_chunks, G), jxs::duo_words_per_env(G), jxs::rk4_lds_words_per_env(G, pk.P.n_chunks)}), limit);
    struct OobCheck {
      const jxs::HostLanes<T, G>& l;
      ~OobCheck() {
        if (l.lds_oob_ >= 0) g_err = "LDS access at word " + std::to_string(l.lds_oob_) + " beyond the " + std::to_string(l.lds_limit_) + " words this launch allocates per environment";
      }
    } oob_check{ln};
    jxs::Core<jxs::HostLanes<T, G>> core(pk.P, a, ln);
    if (mode == (jxs::MODE_STEP | 0x100)) {
      // two-wave workgroup: the inertia wave, then the main wave, on the same LDS image
      core.run_inertia();
      core.template run<jxs::MODE_STEP, jxs::ROLE_MAIN>();
      if (ln.flag_error_) g_err = "two-wave protocol: the main wave waited for a level the inertia wave had not published";
      continue;
    }
    switch (mode) {
      case jxs::MODE_STEP: core.template run<jxs::MODE_STEP>(); break;
      case jxs::MODE_FD: core.template run<jxs::MODE_FD>(); break;
      case jxs::MODE_ID: core.template run<jxs::MODE_ID>(); break;
      case jxs::MODE_ROLLOUT: core.template run<jxs::MODE_ROLLOUT>(); break;
      case jxs::MODE_STEP_RK4: core.template run<jxs::MODE_STEP_RK4>(); break;
      case jxs::MODE_STEP_RIGID: core.template run<jxs::MODE_STEP_RIGID>(); break;
      case jxs::MODE_STEP_RK4_RIGID: core.template run<jxs::MODE_STEP_RK4_RIGID>(); break;
      case jxs::MODE_CRBA: core.template run<jxs::MODE_CRBA>(); break;
      case jxs::MODE_JAC: core.template run<jxs::MODE_JAC>(); break;
      case jxs::MODE_MINV: core.template run<jxs::MODE_MINV>(); break;
      case jxs::MODE_GRAV: core.template run<jxs::MODE_GRAV>(); break;
      case jxs::MODE_DYN: core.template run<jxs::MODE_DYN>(); break;
      case jxs::MODE_DYN_RIGID: core.template run<jxs::MODE_DYN_RIGID>(); break;
      default: core.template run<jxs::MODE_KIN>(); break;
    }
  }
}
template void run_group<JXS_EMUL_UNIT_T, JXS_EMUL_UNIT_G>(const jxs::Packed<JXS_EMUL_UNIT_T>&, jxs::KArgs<JXS_EMUL_UNIT_T>, int);
}  // namespace jxs_emul
#else

template <typename T>
int run_typed(const jxs_model_desc* d, int mode, const void* state_in, void* state_out, const void* tau,
              const void* link_f, int force_repr, const void* in_a, void* out_a, void* out_H, void* out_V,
              int N, int n_steps) {
  jxs::Packed<T> pk;
  const std::string err = jxs::pack_model<T>(*d, pk);
  if (!err.empty()) {
    g_err = err;
    return JXS_EINVAL;
  }
  jxs::KArgs<T> a{};
  a.state_in = static_cast<const T*>(state_in);
  a.state_out = static_cast<T*>(state_out);
  a.tau = (pk.P.n > 0) ? static_cast<const T*>(tau) : nullptr;
  a.link_f = static_cast<const T*>(link_f);
  a.force_repr = force_repr;
  a.in_a = static_cast<const T*>(in_a);
  a.out_a = static_cast<T*>(out_a);
  a.out_H = static_cast<T*>(out_H);
  a.out_V = static_cast<T*>(out_V);
  a.N = N;
  a.n_steps = n_steps;
  a.dbg = static_cast<long long*>(g_dbg_ptr);
  a.has_lds = 1;  // (the host lanes always carry an LDS image: the staged base-state loads are exercised)
  if ((mode & 0xff) == jxs::MODE_STEP && state_out != state_in && pk.n_disabled > 0)
{
    const int tile = 64 / pk.G;
    std::memcpy(state_out, state_in, sizeof(T) * (size_t)((N + tile - 1) / tile) * tile * pk.P.n_rows);
  }
  int launches = 1;
  // bit 9 of the mode: `tau` is a sequence [n_steps * n][N], one block of rows per step (jxs_rollout_controlled)
  const bool tau_seq = (mode & 0x200) != 0 && a.tau != nullptr;
  // bit 10: recorded rollout -- `out_a` is the trajectory block [n_steps * n_rows][N] (jxs_rollout_recorded)
  T* const traj = (mode & 0x400) != 0 ? static_cast<T*>(out_a) : nullptr;
  if (traj != nullptr) a.out_a = nullptr;
  mode &= ~0x600;
  const bool duo = (mode == (jxs::MODE_STEP | 0x100));  // emulate the two-wave workgroup variant of the step kernel
  if (duo) {
    mode = jxs::MODE_STEP;
    if (!(pk.P.row_mode == 1 && pk.P.rigid == 0 && pk.P.n_chunks <= 1 && pk.integrator != JXS_INTEGRATOR_RUNGE_KUTTA4 && pk.G >= 8)) {
      g_err = "the two-wave variant does not apply to this model";
      return JXS_EINVAL;
    }
  }
  if (mode == jxs::MODE_DYN) {  // like the library (jxs_api.hip run_typed)
    if (pk.P.rigid) mode = jxs::MODE_DYN_RIGID;
    a.fparam = T(1);  // (the harness has no argument for the Baumgarte gain: the reference's default)
    if (pk.P.n_chunks == 0) a.out_H = nullptr;  // (the caller's buffer is zero-initialised)
  }
  const bool rk4 = (mode == jxs::MODE_STEP && pk.integrator == JXS_INTEGRATOR_RUNGE_KUTTA4);
  if (rk4) mode = pk.P.rigid ? jxs::MODE_STEP_RK4_RIGID : jxs::MODE_STEP_RK4;
  const bool rigid = (mode == jxs::MODE_STEP && pk.P.rigid);
  if (rigid) mode = jxs::MODE_STEP_RIGID;
  if ((rk4 || rigid) && n_steps > 1) {
    launches = n_steps;
    a.n_steps = 1;
  }
  if (mode == jxs::MODE_STEP && n_steps > 1 && pk.P.n_chunks > 1) {  // like jxs_rollout: not fused
    launches = n_steps;
    a.n_steps = 1;
  }
  if (mode == jxs::MODE_STEP && a.n_steps > 1 && !duo) mode = jxs::MODE_ROLLOUT;
  if (duo) {
    launches = n_steps;
    a.n_steps = 1;
    mode = jxs::MODE_STEP | 0x100;
  }
  g_err.clear();
  const T* const tau_all = a.tau;
  std::vector<T> tau_step;
  if (tau_seq && mode == jxs::MODE_ROLLOUT) a.flags |= 4;
  if (traj != nullptr && mode == jxs::MODE_ROLLOUT && pk.n_disabled > 0) {  // like the library: not fused
    launches = n_steps, a.n_steps = 1, mode = jxs::MODE_STEP;
    if (tau_seq) a.flags &= ~4;
  }
  if (traj != nullptr && mode == jxs::MODE_ROLLOUT) a.out_a = traj;
  for (int it = 0; it < launches; ++it) {
    if (it == 1) a.state_in = a.state_out;
    if (tau_seq && mode != jxs::MODE_ROLLOUT) {  // one launch per step: the step's rows of every tile, like the library's strided copy
      const int tile = 64 / pk.G, n = pk.P.n, tiles = (N + tile - 1) / tile;
      tau_step.resize((size_t)tiles * n * tile);
      for (int t = 0; t < tiles; ++t)
        std::memcpy(&tau_step[(size_t)t * n * tile], tau_all + ((size_t)t * n_steps + it) * n * tile, sizeof(T) * (size_t)n * tile);
      a.tau = tau_step.data();
    }
  switch (pk.G) {
    case 4: run_group<T, 4>(pk, a, mode); break;
    case 8: run_group<T, 8>(pk, a, mode); break;
    case 16: run_group<T, 16>(pk, a, mode); break;
    case 32: run_group<T, 32>(pk, a, mode); break;
    case 64: run_group<T, 64>(pk, a, mode); break;
    default: g_err = "bad group size"; return JXS_EINVAL;
  }
    if (traj != nullptr && mode != jxs::MODE_ROLLOUT) {  // one launch per step: copy the state block into its rows of every tile
      const int tile = 64 / pk.G, rows = pk.P.n_rows, tiles = (N + tile - 1) / tile;
      for (int t = 0; t < tiles; ++t)
        std::memcpy(traj + ((size_t)t * n_steps + it) * rows * tile, a.state_out + (size_t)t * rows * tile, sizeof(T) * (size_t)rows * tile);
    }
  }
  if (!g_err.empty()) return JXS_EINVAL;
  return JXS_OK;
}

}  // namespace jxs_emul
using namespace jxs_emul;

template <typename T>
static int layout_typed(const jxs_model_desc* d, jxs_layout* out) {
  jxs::Packed<T> pk;
  const std::string err = jxs::pack_model<T>(*d, pk);
  if (!err.empty()) {
    g_err = err;
    return JXS_EINVAL;
  }
  const auto& P = pk.P;
  *out = jxs_layout{P.nL, P.n, P.n_points, P.n_rows, P.row_pos, P.row_quat, P.row_s,
                    P.row_vlin, P.row_vang, P.row_sd, P.row_m, pk.G, 64 / pk.G, d->dtype, P.row_mode};
  return JXS_OK;
}

extern "C" {

const char* jxs_emul_last_error(void) { return g_err.c_str(); }
void jxs_emul_set_debug(void* p) { g_dbg_ptr = p; }

int jxs_emul_layout(const jxs_model_desc* d, jxs_layout* out) {
  // (the precision matters: some limits -- the LDS budget of the rigid contact models -- depend on it)
  return d->dtype == JXS_F64 ? layout_typed<double>(d, out) : layout_typed<float>(d, out);
}

int jxs_emul_run(const jxs_model_desc* d, int mode, const void* state_in, void* state_out, const void* tau,
                 const void* link_f, int force_repr, const void* in_a, void* out_a, void* out_H, void* out_V,
                 int N, int n_steps) {
  if (d->dtype == JXS_F64)
    return run_typed<double>(d, mode, state_in, state_out, tau, link_f, force_repr, in_a, out_a, out_H, out_V, N, n_steps);
  return run_typed<float>(d, mode, state_in, state_out, tau, link_f, force_repr, in_a, out_a, out_H, out_V, N, n_steps);
}
}
#endif  // main unit
