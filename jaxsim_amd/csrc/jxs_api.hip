// C-ABI implementation (include/jaxsim_amd.h): gfx950 kernels + launchers + device plumbing.
//
// Launch geometry: one wavefront per workgroup, G lanes per environment, 64/G environments
// per wave.  At the benchmark size (N = 1024 environments per GPU, G = 32) that is 512
// single-wave workgroups, i.e. two waves on each of the 256 CUs, placed by the dispatcher on
// different SIMDs; all cross-lane traffic is wavefront shuffles, no LDS allocation, no
// barriers.
#include <cstdint>
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>

#include "../../include/jaxsim_amd.h"
#include "jxs_params.h"
#include "jxs_pack.h"

// kernels and launchers live in jxs_inst.hip (one translation unit per dtype and mode, jxs_kernels.h)
namespace jxs_launch {
template <typename T, int MODE>
hipError_t launch_g(int G, const jxs::KParams<T>& P, const unsigned char* mblk, const jxs::KArgs<T>& A, hipStream_t s);
}
using jxs_launch::launch_g;

namespace {
// Developer knobs of the launcher (JXS_NO_MFMA, JXS_DUO, JXS_DUO_MAX_BLOCKS, JXS_DISABLE_COMMON_VARIANT): read from the
// environment ONCE per process -- getenv on every launch of a ~6 us kernel was host time on the hot path and a data race
// with a Python thread that changes os.environ (glibc getenv against setenv).  jxs_debug_reload_env() re-reads them: the
// tests switch variants between launches of one process.  Launches captured in a hipGraph keep what they captured.
std::atomic<int> g_knobs{-1}, g_duo_max_blocks{0};
void load_knobs() {
  auto on = [](const char* name) { const char* v = std::getenv(name); return v != nullptr && std::atoi(v) != 0; };
  int k = 0;
  if (on("JXS_NO_MFMA")) k |= jxs::KNOB_NO_MFMA;
  if (on("JXS_DUO")) k |= jxs::KNOB_DUO;
  if (std::getenv("JXS_DISABLE_COMMON_VARIANT") != nullptr) k |= jxs::KNOB_NO_COMMON_VARIANT;
  const char* b = std::getenv("JXS_DUO_MAX_BLOCKS");
  g_duo_max_blocks.store(b == nullptr ? 0 : std::atoi(b));
  g_knobs.store(k);
}
void debug_knobs(int& knobs, int& duo_max_blocks) {
  static std::once_flag once;
  std::call_once(once, load_knobs);
  knobs = g_knobs.load(std::memory_order_relaxed), duo_max_blocks = g_duo_max_blocks.load(std::memory_order_relaxed);
}
}  // namespace

namespace {

thread_local std::string g_err;
long long* g_dbg = nullptr;  // developer profiling build only (jxs_debug_set_stamp_buffer)

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
int hip_fail(hipError_t e, const char* what) {
  return fail(JXS_ENODEVICE, std::string(what) + ": " + hipGetErrorString(e));
}
#define JXS_HIP(call)                                  \
  do {                                                 \
    hipError_t e_ = (call);                            \
    if (e_ != hipSuccess) return hip_fail(e_, #call);  \
  } while (0)

template <typename T>
hipError_t launch_mode(int mode, int G, const jxs::KParams<T>& P, const unsigned char* mblk, const jxs::KArgs<T>& A, hipStream_t s) {
  switch (mode) {
    case jxs::MODE_STEP: return launch_g<T, jxs::MODE_STEP>(G, P, mblk, A, s);
    case jxs::MODE_FD: return launch_g<T, jxs::MODE_FD>(G, P, mblk, A, s);
    case jxs::MODE_ID: return launch_g<T, jxs::MODE_ID>(G, P, mblk, A, s);
    case jxs::MODE_ROLLOUT: return launch_g<T, jxs::MODE_ROLLOUT>(G, P, mblk, A, s);
    case jxs::MODE_STEP_RK4: return launch_g<T, jxs::MODE_STEP_RK4>(G, P, mblk, A, s);
    case jxs::MODE_STEP_RIGID: return launch_g<T, jxs::MODE_STEP_RIGID>(G, P, mblk, A, s);
    case jxs::MODE_STEP_RK4_RIGID: return launch_g<T, jxs::MODE_STEP_RK4_RIGID>(G, P, mblk, A, s);
    case jxs::MODE_CRBA: return launch_g<T, jxs::MODE_CRBA>(G, P, mblk, A, s);
    case jxs::MODE_JAC: return launch_g<T, jxs::MODE_JAC>(G, P, mblk, A, s);
    case jxs::MODE_MINV: return launch_g<T, jxs::MODE_MINV>(G, P, mblk, A, s);
    case jxs::MODE_GRAV: return launch_g<T, jxs::MODE_GRAV>(G, P, mblk, A, s);
    case jxs::MODE_DYN: return launch_g<T, jxs::MODE_DYN>(G, P, mblk, A, s);
    case jxs::MODE_DYN_RIGID: return launch_g<T, jxs::MODE_DYN_RIGID>(G, P, mblk, A, s);
    default: return launch_g<T, jxs::MODE_KIN>(G, P, mblk, A, s);
  }
}

// Host tables of one model + its device model block (jxs_params.h: KParams | ltf | lti | rti | point chunks)
template <typename T>
struct ModelT {
  jxs::Packed<T> pk;
  unsigned char* mblk = nullptr;
  // model-specialised kernels (jxs_model_attach_specialized): launch entry per mode, null = generic kernel
  using SpecLaunch = int (*)(const void*, const unsigned char*, const void*, void*);
  SpecLaunch spec_launch[16] = {};
  void* spec_handle[16] = {};
  int* faults = nullptr;  // [2] discarded contact-force / impact solves since the last reset (rigid contact models)
  // jxs_rollout_controlled, one launch per step: the torques of the current step, [n][N] -- one block per stream
  std::mutex scratch_mu;
  std::map<hipStream_t, std::pair<void*, size_t>> tau_scratch;

  ~ModelT() {
    // The specialised-kernel objects are NOT unloaded (attach_typed): launches of this model may still be in
    // flight on some stream, and unloading a code object under a running kernel is undefined; hipFree
    // synchronises the device.  The objects stay mapped for the life of the process (a few hundred KB each),
    // which also keeps them visible in /proc/self/maps to whoever audits which native code ran.
    (void)hipFree(mblk);
    (void)hipFree(faults);
    for (auto& kv : tau_scratch)
      if (kv.second.first != nullptr) (void)hipFree(kv.second.first);
  }

  hipError_t upload_all() {
    const std::vector<unsigned char> b = pk.block();
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&mblk), b.size());
    if (e != hipSuccess) return e;
    if ((e = hipMemcpy(mblk, b.data(), b.size(), hipMemcpyHostToDevice)) != hipSuccess) return e;
    if ((e = hipMalloc(reinterpret_cast<void**>(&faults), 2 * sizeof(int))) != hipSuccess) return e;
    return hipMemset(faults, 0, 2 * sizeof(int));
  }
  jxs::KArgs<T> args(int N) const {
    jxs::KArgs<T> a{};
    a.N = N;
    a.faults = faults;
    debug_knobs(a.knobs, a.duo_max_blocks);
    return a;
  }
};

}  // namespace

struct jxs_model {
  unsigned long long uid = 0;  // unique per created model: cached launch graphs are keyed by it, not by the address
  int dtype = JXS_F32;
  std::unique_ptr<ModelT<float>> f32;
  std::unique_ptr<ModelT<double>> f64;
};

namespace {

template <typename T>
ModelT<T>* typed(jxs_model* m);
template <>
ModelT<float>* typed<float>(jxs_model* m) { return m->f32.get(); }
template <>
ModelT<double>* typed<double>(jxs_model* m) { return m->f64.get(); }

template <typename T>
int create_typed(const jxs_model_desc* d, std::unique_ptr<ModelT<T>>& slot) {
  auto mt = std::make_unique<ModelT<T>>();
  const std::string err = jxs::pack_model<T>(*d, mt->pk);
  if (!err.empty()) return fail(JXS_EINVAL, err);
  hipError_t e = mt->upload_all();
  if (e != hipSuccess) return hip_fail(e, "uploading model tables");
  slot = std::move(mt);
  return JXS_OK;
}

template <typename T>
int run_typed(jxs_model* model, int mode, const void* state_in, void* state_out, const void* tau,
              const void* link_f, int force_repr, const void* in_a, void* out_a, void* out_H, void* out_V, int N,
              int repeat, void* stream, void* out_tau, bool fuse, int extra_flags, void* traj = nullptr, double fparam = 0.0) {
  ModelT<T>* mt = typed<T>(model);
  hipStream_t s = static_cast<hipStream_t>(stream);
  jxs::KArgs<T> a = mt->args(N);
  a.state_in = static_cast<const T*>(state_in);
  a.state_out = static_cast<T*>(state_out);
  a.tau = (mt->pk.P.n > 0) ? static_cast<const T*>(tau) : nullptr;  // no joints: nothing to read
  a.link_f = static_cast<const T*>(link_f);
  a.force_repr = force_repr;
  a.in_a = static_cast<const T*>(in_a);
  a.out_a = static_cast<T*>(out_a);
  a.out_H = static_cast<T*>(out_H);
  a.out_V = static_cast<T*>(out_V);
  if (mode == jxs::MODE_STEP && state_out != state_in &&
      (mt->pk.n_disabled > 0 || (mt->pk.P.rigid && mt->pk.P.n_points > 0))) {
    // rows of disabled collidable points are not touched by the kernel, and the rigid contact models
    // have no tangential deformation (their rows are passengers of the state block): carry them over
    const int tile = 64 / mt->pk.G;
    const size_t elems = (size_t)((N + tile - 1) / tile) * tile * mt->pk.P.n_rows;
    JXS_HIP(hipMemcpyAsync(state_out, state_in, sizeof(T) * elems, hipMemcpyDeviceToDevice, s));
  }
  if (mode == jxs::MODE_DYN) {
    if (mt->pk.P.rigid) mode = jxs::MODE_DYN_RIGID;
    a.fparam = static_cast<T>(fparam);
    // rows the kernel does not write: the deformation rates of disabled points, and of every point of the rigid contact
    // models (no tangential state: their rows are passengers of the block) -- zero, like the reference's m_dot
    if (state_out != nullptr && (mt->pk.n_disabled > 0 || (mt->pk.P.rigid && mt->pk.P.n_points > 0))) {
      const int tile = 64 / mt->pk.G;
      const size_t elems = (size_t)((N + tile - 1) / tile) * tile * mt->pk.P.n_rows;
      JXS_HIP(hipMemsetAsync(state_out, 0, sizeof(T) * elems, s));
    }
    if (out_H != nullptr && mt->pk.P.n_chunks == 0) {  // no enabled points: the kernel has no contact phase
      const int tile = 64 / mt->pk.G;
      JXS_HIP(hipMemsetAsync(out_H, 0, sizeof(T) * (size_t)((N + tile - 1) / tile) * tile * mt->pk.P.nL * 6, s));
      a.out_H = nullptr;
      if (state_out == nullptr) return JXS_OK;
    }
  }
  a.dbg = g_dbg;
  a.flags |= extra_flags;
  a.out_tau = static_cast<T*>(out_tau);  // jxs_gravity_torques: RNEA at zero velocity, joint torques only
  a.id_zero_vel = out_tau != nullptr ? 1 : 0;
  a.n_steps = 1;
  if (mode == jxs::MODE_STEP && mt->pk.integrator == JXS_INTEGRATOR_RUNGE_KUTTA4) {
    // four dynamics evaluations per launch; a rollout is one launch per step
    mode = mt->pk.P.rigid ? jxs::MODE_STEP_RK4_RIGID : jxs::MODE_STEP_RK4;
  }
  if (mode == jxs::MODE_STEP && mt->pk.P.rigid) mode = jxs::MODE_STEP_RIGID;  // QP contacts + impact, one launch per step
  // (a recorded rollout of a model with disabled collidable points is not fused: their rows are not written by the kernel)
  // [round 4] ... and a plain rollout is fused only while the grid is at most eight waves per SIMD: beyond that the
  // per-step launches are FASTER (measured, tools/sweep.py --rollout, humanoid: 8.5 against 10.2 us per step at 4096
  // environments, 32.3 = 32.3 at 16384, 128.7 against 109.2 at 65536 -- the long-lived waves of the fused launch fill
  // the chip in fewer, coarser rounds and the prologue it saves is hidden by the other waves anyway)
  static const int n_simds = [] {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return 4 * (cus > 0 ? cus : 256);
  }();
  const int blocks = (N + (64 / mt->pk.G) - 1) / (64 / mt->pk.G);
  const bool worth_fusing = traj != nullptr || blocks <= 8 * n_simds;
  if (fuse && worth_fusing && mode == jxs::MODE_STEP && repeat > 1 && mt->pk.P.n_chunks <= 1 && (traj == nullptr || mt->pk.n_disabled == 0)) {
    // fused rollout: one launch, the state stays in registers between the steps
    a.n_steps = repeat;
    repeat = 1;
    mode = jxs::MODE_ROLLOUT;
    if (traj != nullptr) a.out_a = static_cast<T*>(traj);  // recorded: the kernel stores the state after every step (jxs_core.h)
  }
  const bool tau_seq = (extra_flags & 4) != 0 && a.tau != nullptr;
  void* scratch = nullptr;  // the torques of the current step, [n][N] (one launch per step)
  const T* const tau_all = a.tau;
  const int seq_steps = a.n_steps > 1 ? a.n_steps : repeat;
  if (tau_seq && mode != jxs::MODE_ROLLOUT) {
    // one launch per step (RungeKutta4, the rigid contact models, several point chunks): the torques of step `it` are
    // rows it * n ... of every tile of the sequence [n_steps * n][N] -- gathered into a [n][N] block per step (a strided
    // device copy), which the step kernel reads like any joint_force_references
    a.flags &= ~4;
    const int tile = 64 / mt->pk.G;
    const size_t need = sizeof(T) * (size_t)((N + tile - 1) / tile) * tile * mt->pk.P.n;
    // [ADVICE r4] one gather block per (model, STREAM), under a mutex: two streams or threads that roll out the same model
    // never share a block (round 4 kept one per model), and the launch path allocates only when a stream's block has to
    // grow -- which a stream capture cannot contain: refused there.  (hipMallocAsync / hipFreeAsync per call was tried
    // first and returned wrong rollouts on ROCm 7.2: the strided device copy into pool memory did not land.)
    std::lock_guard<std::mutex> lk(mt->scratch_mu);
    auto& blk = mt->tau_scratch[s];
    if (blk.second < need) {
      hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
      if (hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
        return fail(JXS_EINVAL, "a rollout with a torque sequence that runs one launch per step has to (re)allocate its gather block: not inside a stream capture");
      if (blk.first != nullptr) {
        JXS_HIP(hipStreamSynchronize(s));  // the last launches of this stream may still read the old block
        (void)hipFree(blk.first);
        blk = {nullptr, 0};
      }
      JXS_HIP(hipMalloc(&blk.first, need));
      blk.second = need;
    }
    scratch = blk.first;
  }
  for (int it = 0; it < repeat; ++it) {
    if (tau_seq && mode != jxs::MODE_ROLLOUT) {
      const int tile = 64 / mt->pk.G, n = mt->pk.P.n;
      const size_t row_bytes = sizeof(T) * (size_t)tile;
      JXS_HIP(hipMemcpy2DAsync(scratch, row_bytes * n, reinterpret_cast<const char*>(tau_all) + row_bytes * n * it,
                               row_bytes * n * seq_steps, row_bytes * n, (size_t)((N + tile - 1) / tile), hipMemcpyDeviceToDevice, s));
      a.tau = static_cast<const T*>(scratch);
    }
    hipError_t e = (mode >= 0 && mode < 16 && mt->spec_launch[mode] != nullptr)
                       ? static_cast<hipError_t>(mt->spec_launch[mode](&mt->pk.P, mt->mblk, &a, s))
                       : launch_mode<T>(mode, mt->pk.G, mt->pk.P, mt->mblk, a, s);
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    if (traj != nullptr && mode != jxs::MODE_ROLLOUT) {
      // recorded rollout, one launch per step: the state block after step `it` goes to rows it * n_rows ... of every tile
      const int tile = 64 / mt->pk.G, rows = mt->pk.P.n_rows;
      const size_t blk_bytes = sizeof(T) * (size_t)tile * rows;
      JXS_HIP(hipMemcpy2DAsync(static_cast<char*>(traj) + blk_bytes * it, blk_bytes * seq_steps, a.state_out, blk_bytes, blk_bytes,
                               (size_t)((N + tile - 1) / tile), hipMemcpyDeviceToDevice, s));
    }
    if (it == 0 && repeat > 1) a.state_in = a.state_out;  // unfused rollout continues from its own output
  }
  return JXS_OK;
}

int run_any(jxs_model* model, int mode, const void* state_in, void* state_out, const void* tau, const void* link_f,
            int force_repr, const void* in_a, void* out_a, void* out_H, void* out_V, int N, int repeat,
            void* stream, void* out_tau = nullptr, bool fuse = true, int extra_flags = 0, void* traj = nullptr, double fparam = 0.0) {
  if (model == nullptr) return fail(JXS_EINVAL, "null model");
  if (state_in == nullptr) return fail(JXS_EINVAL, "null state");
  if (N <= 0) return fail(JXS_EINVAL, "N must be positive");
  if (force_repr < 0 || force_repr > 2) return fail(JXS_EINVAL, "invalid force representation");
  if (model->dtype == JXS_F64)
    return run_typed<double>(model, mode, state_in, state_out, tau, link_f, force_repr, in_a, out_a, out_H, out_V, N,
                             repeat, stream, out_tau, fuse, extra_flags, traj, fparam);
  return run_typed<float>(model, mode, state_in, state_out, tau, link_f, force_repr, in_a, out_a, out_H, out_V, N,
                          repeat, stream, out_tau, fuse, extra_flags, traj, fparam);
}

// ---- RCCL, resolved lazily so that the library loads (and the CPU symbol test passes)
// ---- without pulling a second RCCL next to the one a host launcher may already hold.
struct Rccl {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;

int rccl_load() {
  if (g_rccl.h != nullptr) return JXS_OK;
  void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (h == nullptr) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (h == nullptr) return fail(JXS_ECOMM, std::string("cannot load RCCL: ") + dlerror());
  Rccl r;
  r.h = h;
  r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
  r.GetVersion = reinterpret_cast<decltype(r.GetVersion)>(dlsym(h, "ncclGetVersion"));  // (optional)
  r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
  r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
  r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(h, "ncclAllGather"));
  r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
  if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.GetErrorString)
    return fail(JXS_ECOMM, "RCCL library lacks an expected symbol");
  g_rccl = r;
  return JXS_OK;
}
int rccl_fail(ncclResult_t r, const char* what) {
  return fail(JXS_ECOMM, std::string(what) + ": " + g_rccl.GetErrorString(r));
}

}  // namespace

// ---- value checks standing in for JAXSIM_ENABLE_EXCEPTIONS (src/jaxsim/rbda/utils.py:135-146) ----
template <typename T>
__global__ void jxs_validate_kernel(const T* state, int n_rows, int row_quat, int tile, int N, int* counts) {
  const int env = blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= N) return;
  const size_t base = (size_t)(env / tile) * n_rows * tile + env % tile;
  T q2 = 0;
  bool nan_q = false;
  for (int k = 0; k < 4; ++k) {
    const T v = state[base + (size_t)(row_quat + k) * tile];
    nan_q = nan_q || (v != v);
    q2 += v * v;
  }
  bool bad = false;
  for (int r = 0; r < n_rows; ++r) {
    const T v = state[base + (size_t)r * tile];
    bad = bad || !(v - v == T(0));  // NaN or infinity
  }
  const T d = q2 > T(1) ? q2 - T(1) : T(1) - q2;
  if (nan_q) atomicAdd(&counts[0], 1);
  if (!nan_q && !(d <= T(1e-8) + T(1e-5))) atomicAdd(&counts[1], 1);  // jnp.allclose(q.q, 1.0)
  if (bad) atomicAdd(&counts[2], 1);
}


// ---- layout conversion at the boundary: environment-major [N][rows] (what a host array, a DLPack /
// __cuda_array_interface__ buffer of another framework holds) <-> the tile-interleaved storage ------
template <typename T, bool TO_TILED>
__global__ void jxs_retile_kernel(const T* src, T* dst, int rows, int N, int tile) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)rows * N) return;
  const int env = (int)(i / rows), row = (int)(i % rows);
  const size_t tiled = ((size_t)(env / tile) * rows + row) * tile + env % tile;
  if (TO_TILED) dst[tiled] = src[i];
  else dst[i] = src[tiled];
}

namespace {
template <typename T>
int attach_typed(jxs_model* model, ModelT<T>* mt, int mode, const char* path) {
  const std::string want = jxs::kernel_spec_string(mt->pk, mode);
  void* h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (h == nullptr) return fail(JXS_EINVAL, std::string("cannot load the specialised kernel: ") + dlerror());
  auto text = reinterpret_cast<const char* (*)()>(dlsym(h, "jxs_spec_string"));
  auto launch = reinterpret_cast<typename ModelT<T>::SpecLaunch>(dlsym(h, "jxs_spec_launch"));
  if (text == nullptr || launch == nullptr) {
    dlclose(h);
    return fail(JXS_EINVAL, "not a specialised-kernel object (jxs_spec_string / jxs_spec_launch missing)");
  }
  if (want != text()) {
    const std::string got = text();
    dlclose(h);
    return fail(JXS_EINVAL, "the specialised kernel was built for another model: " + got + " != " + want);
  }
  mt->spec_handle[mode] = h;  // (a replaced object stays loaded, see ~ModelT; dlopen of the same path returns the same handle)
  mt->spec_launch[mode] = launch;
  static std::atomic<unsigned long long> next_uid{1ull << 40};
  model->uid = next_uid.fetch_add(1);  // launch graphs captured with the generic kernel are not found again
  return JXS_OK;
}
}  // namespace

extern "C" {

// Not part of the public ABI: phase-stamp buffer of the -DJXS_PHASE_TIMING developer build.
int jxs_debug_set_stamp_buffer(void* dptr) {
  g_dbg = static_cast<long long*>(dptr);
  return JXS_OK;
}

const char* jxs_last_error(void) { return g_err.c_str(); }

int jxs_device_count(int* count) {
  if (count == nullptr) return fail(JXS_EINVAL, "null count");
  hipError_t e = hipGetDeviceCount(count);
  if (e != hipSuccess) {
    *count = 0;
    return hip_fail(e, "hipGetDeviceCount");
  }
  return JXS_OK;
}
int jxs_set_device(int device) {
  JXS_HIP(hipSetDevice(device));
  return JXS_OK;
}
int jxs_malloc(void** dptr, uint64_t bytes) {
  if (dptr == nullptr) return fail(JXS_EINVAL, "null dptr");
  hipError_t e = hipMalloc(dptr, bytes > 0 ? bytes : 1);
  if (e == hipErrorOutOfMemory) return fail(JXS_ENOMEM, "hipMalloc: out of memory");
  if (e != hipSuccess) return hip_fail(e, "hipMalloc");
  return JXS_OK;
}
int jxs_free(void* dptr) {
  JXS_HIP(hipFree(dptr));
  return JXS_OK;
}
int jxs_memcpy_h2d(void* dst, const void* src, uint64_t bytes, void* stream) {
  JXS_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, static_cast<hipStream_t>(stream)));
  JXS_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
  return JXS_OK;
}
int jxs_memcpy_d2h(void* dst, const void* src, uint64_t bytes, void* stream) {
  JXS_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream)));
  JXS_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
  return JXS_OK;
}
int jxs_memcpy_d2d(void* dst, const void* src, uint64_t bytes, void* stream) {
  JXS_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)));
  return JXS_OK;
}
int jxs_memset(void* dst, int value, uint64_t bytes, void* stream) {
  JXS_HIP(hipMemsetAsync(dst, value, bytes, static_cast<hipStream_t>(stream)));
  return JXS_OK;
}
int jxs_stream_create(void** stream) {
  if (stream == nullptr) return fail(JXS_EINVAL, "null stream");
  hipStream_t s;
  JXS_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  *stream = s;
  return JXS_OK;
}
int jxs_stream_destroy(void* stream) {
  JXS_HIP(hipStreamDestroy(static_cast<hipStream_t>(stream)));
  return JXS_OK;
}
int jxs_stream_synchronize(void* stream) {
  JXS_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
  return JXS_OK;
}
int jxs_stream_wait_spin(void* stream) {
  // busy-wait on the stream from the calling thread: the completion is seen within a microsecond or two
  // instead of after the wake-up of a blocking wait (bench.py brackets short timed regions with it)
  hipStream_t s = static_cast<hipStream_t>(stream);
  for (;;) {
    const hipError_t e = hipStreamQuery(s);
    if (e == hipSuccess) return JXS_OK;
    if (e != hipErrorNotReady) return hip_fail(e, "hipStreamQuery");
  }
}
int jxs_device_synchronize(void) {
  JXS_HIP(hipDeviceSynchronize());
  return JXS_OK;
}
int jxs_event_create(void** event) {
  if (event == nullptr) return fail(JXS_EINVAL, "null event");
  hipEvent_t e;
  JXS_HIP(hipEventCreate(&e));
  *event = e;
  return JXS_OK;
}
int jxs_event_destroy(void* event) {
  JXS_HIP(hipEventDestroy(static_cast<hipEvent_t>(event)));
  return JXS_OK;
}
int jxs_event_record(void* event, void* stream) {
  JXS_HIP(hipEventRecord(static_cast<hipEvent_t>(event), static_cast<hipStream_t>(stream)));
  return JXS_OK;
}
int jxs_event_elapsed_ms(void* start, void* stop, float* ms) {
  JXS_HIP(hipEventSynchronize(static_cast<hipEvent_t>(stop)));
  JXS_HIP(hipEventElapsedTime(ms, static_cast<hipEvent_t>(start), static_cast<hipEvent_t>(stop)));
  return JXS_OK;
}

int jxs_model_create(const jxs_model_desc* desc, jxs_model** out) {
  if (desc == nullptr || out == nullptr) return fail(JXS_EINVAL, "null argument");
  if (desc->dtype != JXS_F32 && desc->dtype != JXS_F64) return fail(JXS_EINVAL, "dtype must be JXS_F32 or JXS_F64");
  if (desc->contact_model == JXS_CONTACT_SOFT && desc->D <= 0 && desc->n_points > 0)
    return fail(JXS_EINVAL, "soft-contact damping D must be positive");
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count == 0)
    return fail(JXS_ENODEVICE, "no HIP device available (this library has no CPU fallback)");
  auto m = std::make_unique<jxs_model>();
  static std::atomic<unsigned long long> next_uid{1};
  m->uid = next_uid.fetch_add(1);
  m->dtype = desc->dtype;
  int rc = (desc->dtype == JXS_F64) ? create_typed<double>(desc, m->f64) : create_typed<float>(desc, m->f32);
  if (rc != JXS_OK) return rc;
  *out = m.release();
  return JXS_OK;
}
int jxs_kernel_spec(const jxs_model_desc* desc, int mode, char* buf, int capacity) {
  if (desc == nullptr || buf == nullptr || capacity <= 0) return fail(JXS_EINVAL, "null argument");
  if (desc->dtype != JXS_F32 && desc->dtype != JXS_F64) return fail(JXS_EINVAL, "dtype must be JXS_F32 or JXS_F64");
  std::string text, err;
  if (desc->dtype == JXS_F64) {
    jxs::Packed<double> pk;
    err = jxs::pack_model<double>(*desc, pk);
    if (err.empty()) text = jxs::kernel_spec_string(pk, mode);
  } else {
    jxs::Packed<float> pk;
    err = jxs::pack_model<float>(*desc, pk);
    if (err.empty()) text = jxs::kernel_spec_string(pk, mode);
  }
  if (!err.empty()) return fail(JXS_EINVAL, err);
  if ((int)text.size() + 1 > capacity) return fail(JXS_EINVAL, "buffer too small for the kernel description");
  std::memcpy(buf, text.c_str(), text.size() + 1);
  return (int)text.size();
}

int jxs_model_attach_specialized(jxs_model* model, int mode, const char* path) {
  if (model == nullptr || path == nullptr) return fail(JXS_EINVAL, "null argument");
  if (mode < 0 || mode >= 16) return fail(JXS_EINVAL, "invalid mode");
  return model->dtype == JXS_F64 ? attach_typed<double>(model, model->f64.get(), mode, path)
                                 : attach_typed<float>(model, model->f32.get(), mode, path);
}
int jxs_model_specialized_modes(const jxs_model* model, unsigned* mask) {
  if (model == nullptr || mask == nullptr) return fail(JXS_EINVAL, "null argument");
  unsigned m = 0;
  for (int k = 0; k < 16; ++k) {
    const bool on = model->dtype == JXS_F64 ? model->f64->spec_launch[k] != nullptr : model->f32->spec_launch[k] != nullptr;
    if (on) m |= 1u << k;
  }
  *mask = m;
  return JXS_OK;
}

int jxs_model_destroy(jxs_model* model) {
  delete model;
  return JXS_OK;
}
int jxs_model_layout(const jxs_model* model, jxs_layout* out) {
  if (model == nullptr || out == nullptr) return fail(JXS_EINVAL, "null argument");
  auto fill = [&](const auto& pk) {
    const auto& P = pk.P;
    *out = jxs_layout{P.nL, P.n, P.n_points, P.n_rows, P.row_pos, P.row_quat, P.row_s,
                      P.row_vlin, P.row_vang, P.row_sd, P.row_m, pk.G, 64 / pk.G, model->dtype, P.row_mode};
  };
  if (model->dtype == JXS_F64) fill(model->f64->pk); else fill(model->f32->pk);
  return JXS_OK;
}

int jxs_validate_state(jxs_model* model, const void* state, int N, int* counts3, void* stream) {
  if (model == nullptr || state == nullptr || counts3 == nullptr) return fail(JXS_EINVAL, "null argument");
  if (N <= 0) return fail(JXS_EINVAL, "N must be positive");
  hipStream_t s = static_cast<hipStream_t>(stream);
  int* d = nullptr;
  JXS_HIP(hipMalloc(&d, 3 * sizeof(int)));
  hipError_t e = hipMemsetAsync(d, 0, 3 * sizeof(int), s);
  if (e != hipSuccess) {
    (void)hipFree(d);
    return hip_fail(e, "jxs_validate_state");
  }
  jxs_layout lay;
  jxs_model_layout(model, &lay);
  const int threads = 256, blocks = (N + threads - 1) / threads;
  if (model->dtype == JXS_F64)
    hipLaunchKernelGGL(jxs_validate_kernel<double>, dim3(blocks), dim3(threads), 0, s, static_cast<const double*>(state),
                       lay.n_rows, lay.row_quat, lay.tile, N, d);
  else
    hipLaunchKernelGGL(jxs_validate_kernel<float>, dim3(blocks), dim3(threads), 0, s, static_cast<const float*>(state),
                       lay.n_rows, lay.row_quat, lay.tile, N, d);
  e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(counts3, d, 3 * sizeof(int), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  (void)hipFree(d);
  if (e != hipSuccess) return hip_fail(e, "jxs_validate_state");
  return JXS_OK;
}

static int retile(const void* src, void* dst, int rows, int N, int tile, int dtype, void* stream, bool to_tiled) {
  if (src == nullptr || dst == nullptr) return fail(JXS_EINVAL, "null argument");
  if (rows <= 0 || N <= 0 || tile <= 0) return fail(JXS_EINVAL, "rows, N and tile must be positive");
  if (dtype != JXS_F32 && dtype != JXS_F64) return fail(JXS_EINVAL, "dtype must be JXS_F32 or JXS_F64");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const long long total = (long long)rows * N;
  const int threads = 256;
  const int blocks = (int)((total + threads - 1) / threads);
  if (dtype == JXS_F64) {
    if (to_tiled) hipLaunchKernelGGL((jxs_retile_kernel<double, true>), dim3(blocks), dim3(threads), 0, s, (const double*)src, (double*)dst, rows, N, tile);
    else hipLaunchKernelGGL((jxs_retile_kernel<double, false>), dim3(blocks), dim3(threads), 0, s, (const double*)src, (double*)dst, rows, N, tile);
  } else {
    if (to_tiled) hipLaunchKernelGGL((jxs_retile_kernel<float, true>), dim3(blocks), dim3(threads), 0, s, (const float*)src, (float*)dst, rows, N, tile);
    else hipLaunchKernelGGL((jxs_retile_kernel<float, false>), dim3(blocks), dim3(threads), 0, s, (const float*)src, (float*)dst, rows, N, tile);
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "jxs_retile");
  return JXS_OK;
}
int jxs_tile_from_env_major(const void* src, void* dst, int rows, int N, int tile, int dtype, void* stream) {
  return retile(src, dst, rows, N, tile, dtype, stream, true);
}
int jxs_tile_to_env_major(const void* src, void* dst, int rows, int N, int tile, int dtype, void* stream) {
  return retile(src, dst, rows, N, tile, dtype, stream, false);
}

int jxs_step(jxs_model* model, const void* state_in, void* state_out, const void* tau, const void* link_forces,
             int force_repr, int N, void* stream) {
  if (state_out == nullptr) return fail(JXS_EINVAL, "null state_out");
  return run_any(model, jxs::MODE_STEP, state_in, state_out, tau, link_forces, force_repr, nullptr, nullptr, nullptr,
                 nullptr, N, 1, stream);
}
int jxs_step_gravity_compensated(jxs_model* model, const void* state_in, void* state_out, const void* tau,
                                 const void* link_forces, int force_repr, int N, void* stream) {
  if (state_out == nullptr) return fail(JXS_EINVAL, "null state_out");
  if (model == nullptr) return fail(JXS_EINVAL, "null model");
  const bool rigid = model->dtype == JXS_F64 ? (model->f64->pk.P.rigid != 0) : (model->f32->pk.P.rigid != 0);
  if (!rigid)
    return fail(JXS_EINVAL, "jxs_step_gravity_compensated: built for the rigid contact models (RigidContacts / RelaxedRigidContacts with "
                            "enabled points); use jxs_gravity_torques + jxs_step");
  return run_any(model, jxs::MODE_STEP, state_in, state_out, tau, link_forces, force_repr, nullptr, nullptr, nullptr, nullptr, N, 1,
                 stream, nullptr, true, 2);
}
int jxs_rollout(jxs_model* model, void* state, const void* tau, const void* link_forces, int force_repr, int N,
                int n_steps, void* stream) {
  if (n_steps < 0) return fail(JXS_EINVAL, "n_steps must be >= 0");
  return run_any(model, jxs::MODE_STEP, state, state, tau, link_forces, force_repr, nullptr, nullptr, nullptr, nullptr,
                 N, n_steps, stream);
}
int jxs_rollout_controlled(jxs_model* model, void* state, const void* tau_seq, const void* link_forces, int force_repr,
                           int N, int n_steps, void* stream) {
  if (n_steps < 0) return fail(JXS_EINVAL, "n_steps must be >= 0");
  if (n_steps == 0) return JXS_OK;
  if (tau_seq == nullptr) return fail(JXS_EINVAL, "null torque sequence (jxs_rollout takes constant or no torques)");
  if (n_steps == 1) return run_any(model, jxs::MODE_STEP, state, state, tau_seq, link_forces, force_repr, nullptr, nullptr, nullptr, nullptr, N, 1, stream);
  return run_any(model, jxs::MODE_STEP, state, state, tau_seq, link_forces, force_repr, nullptr, nullptr, nullptr, nullptr,
                 N, n_steps, stream, nullptr, /*fuse=*/true, /*extra_flags=*/4);
}
int jxs_rollout_recorded(jxs_model* model, void* state, const void* tau, int tau_per_step, const void* link_forces,
                         int force_repr, int N, int n_steps, void* out_states, void* stream) {
  if (n_steps < 0) return fail(JXS_EINVAL, "n_steps must be >= 0");
  if (n_steps == 0) return JXS_OK;
  if (out_states == nullptr) return fail(JXS_EINVAL, "null out_states (jxs_rollout / jxs_rollout_controlled do not record)");
  if (tau_per_step && tau == nullptr) return fail(JXS_EINVAL, "tau_per_step without a torque sequence");
  return run_any(model, jxs::MODE_STEP, state, state, tau, link_forces, force_repr, nullptr, nullptr, nullptr, nullptr,
                 N, n_steps, stream, nullptr, /*fuse=*/true, /*extra_flags=*/(tau_per_step && n_steps > 1) ? 4 : 0, out_states);
}
int jxs_step_repeat(jxs_model* model, void* state, const void* tau, const void* link_forces, int force_repr, int N,
                    int n_launches, void* stream) {
  if (n_launches < 0) return fail(JXS_EINVAL, "n_launches must be >= 0");
  if (n_launches == 0) return JXS_OK;
  if (model == nullptr) return fail(JXS_EINVAL, "null model");
  if (state == nullptr) return fail(JXS_EINVAL, "null state");
  // Blocks of launches are captured once into a hipGraph and replayed: the same kernels in the same
  // order, with less per-launch work on the host and smaller gaps on the device (9.94 -> 9.77 us per step
  // at 1024 humanoids with 50 launches per graph, 9.2 with 250).  Two block sizes: requests of 250 or more
  // launches replay the long graph, then blocks of 50, the rest is launched plainly.  A graph is captured
  // on the first request that can use it (~0.7 ms for the long one).  Needs a created stream (the
  // legacy default stream cannot be captured).
  static const bool use_graph = std::getenv("JXS_DISABLE_STEP_GRAPH") == nullptr;  // developer knob: A/B
  static const int kShort = std::getenv("JXS_STEP_GRAPH_LAUNCHES") ? std::max(2, std::atoi(std::getenv("JXS_STEP_GRAPH_LAUNCHES"))) : 50;
  static const int kLong = std::max(kShort, 250);
  if (use_graph && stream != nullptr && n_launches >= 2) {
    struct Key {
      unsigned long long m; void* st; const void* tau; const void* lf; int repr, N; void* s; int block;
      bool operator==(const Key& o) const {
        return m == o.m && st == o.st && tau == o.tau && lf == o.lf && repr == o.repr && N == o.N && s == o.s && block == o.block;
      }
    };
    struct Slot {
      Key key{};
      hipGraphExec_t exec = nullptr;
    };
    // slot 0: blocks of kLong launches, slot 1: blocks of kShort; the remainder (< kShort) is launched plainly.
    // Measured with the 7.7 us kernel (tools/region_overhead.py, wall time of a region of n launches):
    // graph of exactly n launches 17.1 us + 7.77 n, plain launches 11.0 us + 7.72 n -- a graph launch costs 6 us
    // more to get going than the first of n plain launches, and the host loop in C keeps ahead of the device.
    // (JXS_STEP_GRAPH_REMAINDER=1 brings the graph of exactly the remainder back: slot 2.)
    static const bool graph_remainder = std::getenv("JXS_STEP_GRAPH_REMAINDER") != nullptr;  // developer knob: A/B
    static thread_local Slot slots[3];
    hipStream_t hs = static_cast<hipStream_t>(stream);
    for (int t = (kLong > kShort ? 0 : 1); t < (graph_remainder ? 3 : 2); ++t) {
      const int block = t == 0 ? kLong : t == 1 ? kShort : n_launches;
      if (block < 2 || n_launches < block) continue;
      const Key k{model->uid, state, tau, link_forces, force_repr, N, stream, block};
      Slot& sl = slots[t];
      if (sl.exec == nullptr || !(k == sl.key)) {
        if (sl.exec != nullptr) (void)hipGraphExecDestroy(sl.exec), sl.exec = nullptr;
        hipGraph_t g = nullptr;
        JXS_HIP(hipStreamBeginCapture(hs, hipStreamCaptureModeThreadLocal));
        const int rc = run_any(model, jxs::MODE_STEP, state, state, tau, link_forces, force_repr, nullptr, nullptr, nullptr,
                               nullptr, N, block, stream, nullptr, /*fuse=*/false);
        hipError_t e = hipStreamEndCapture(hs, &g);
        if (rc != JXS_OK) {
          if (g != nullptr) (void)hipGraphDestroy(g);
          return rc;
        }
        if (e != hipSuccess) return hip_fail(e, "hipStreamEndCapture");
        e = hipGraphInstantiate(&sl.exec, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (e != hipSuccess) {
          sl.exec = nullptr;
          return hip_fail(e, "hipGraphInstantiate");
        }
        sl.key = k;
      }
      for (; n_launches >= block; n_launches -= block) JXS_HIP(hipGraphLaunch(sl.exec, hs));
    }
    if (n_launches == 0) return JXS_OK;
  }
  return run_any(model, jxs::MODE_STEP, state, state, tau, link_forces, force_repr, nullptr, nullptr, nullptr, nullptr,
                 N, n_launches, stream, nullptr, /*fuse=*/false);
}
int jxs_step_repeat_timed(jxs_model* model, void* state, const void* tau, const void* link_forces, int force_repr, int N,
                          int n_launches, void* stream, double* seconds) {
  if (seconds == nullptr) return fail(JXS_EINVAL, "null seconds");
  int rc = jxs_stream_wait_spin(stream);  // the region starts on an idle stream ...
  if (rc != JXS_OK) return rc;
  const auto t0 = std::chrono::steady_clock::now();
  rc = jxs_step_repeat(model, state, tau, link_forces, force_repr, N, n_launches, stream);
  if (rc != JXS_OK) return rc;
  // ... and ends when the last launch has completed.  The host learns of it from a word in pinned host memory that the
  // stream writes behind the last launch (hipStreamWriteValue32) and this thread polls -- no runtime call inside the
  // wait (JXS_TIMED_WAIT_QUERY=1: poll hipStreamQuery instead, the round-3 bracket; developer knob, A/B)
  static const bool by_query = std::getenv("JXS_TIMED_WAIT_QUERY") != nullptr;
  // [ADVICE r4] ONE pinned line per process (64 words: a word per calling thread, handed out once and never freed while
  // the library is loaded -- round 4 allocated a line per thread and leaked it), and a poll that cannot hang: a faulted
  // launch or a stream error never writes the word, so every 4096 spins the stream is asked (hipStreamQuery returns the
  // error, or success when the word was missed for another reason) and its status is read once after the word is seen.
  static volatile uint32_t* line = [] {
    void* p = nullptr;
    if (hipHostMalloc(&p, 64 * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) return static_cast<volatile uint32_t*>(nullptr);
    std::memset(p, 0, 64 * sizeof(uint32_t));
    return static_cast<volatile uint32_t*>(p);
  }();
  static std::atomic<int> next_word{0};
  static thread_local int my_word = -1;
  static thread_local uint32_t seq = 0;
  if (my_word < 0) my_word = next_word.fetch_add(1);
  volatile uint32_t* flag = (line != nullptr && my_word < 64) ? line + my_word : nullptr;
  bool polled = false;
  if (!by_query && flag != nullptr && stream != nullptr) {
    const uint32_t want = ++seq;
    hipStream_t hs = static_cast<hipStream_t>(stream);
    if (hipStreamWriteValue32(hs, const_cast<uint32_t*>(flag), want, 0) == hipSuccess) {
      polled = true;
      for (unsigned spins = 0; *flag != want; ++spins) {
        if ((spins & 4095u) == 4095u) {
          const hipError_t q = hipStreamQuery(hs);
          if (q == hipSuccess) break;                                      // the stream drained: the word is (or is about to be) there
          if (q != hipErrorNotReady) return hip_fail(q, "hipStreamQuery");  // a faulted launch: report it instead of spinning for ever
        }
      }
      const hipError_t q = hipStreamQuery(hs);
      if (q != hipSuccess && q != hipErrorNotReady) return hip_fail(q, "hipStreamQuery");
    } else {
      (void)hipGetLastError();
    }
  }
  if (!polled) rc = jxs_stream_wait_spin(stream);
  *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return rc;
}
int jxs_forward_dynamics_aba(jxs_model* model, const void* state, const void* joint_forces, const void* link_forces,
                             int force_repr, void* out_acc, int N, void* stream) {
  if (out_acc == nullptr) return fail(JXS_EINVAL, "null out_acc");
  return run_any(model, jxs::MODE_FD, state, nullptr, joint_forces, link_forces, force_repr, nullptr, out_acc, nullptr,
                 nullptr, N, 1, stream);
}
int jxs_system_dynamics(jxs_model* model, const void* state, const void* joint_torques, const void* link_forces,
                        int force_repr, double baumgarte, void* out_xdot, void* out_link_contact_forces, int N, void* stream) {
  if (out_xdot == nullptr && out_link_contact_forces == nullptr) return fail(JXS_EINVAL, "jxs_system_dynamics: no output asked for");
  if (out_xdot != nullptr && out_xdot == state) return fail(JXS_EINVAL, "jxs_system_dynamics: out_xdot must not alias the state");
  return run_any(model, jxs::MODE_DYN, state, out_xdot, joint_torques, link_forces, force_repr, nullptr, nullptr,
                 out_link_contact_forces, nullptr, N, 1, stream, nullptr, true, 0, nullptr, baumgarte);
}
int jxs_link_contact_forces(jxs_model* model, const void* state, const void* joint_torques, const void* link_forces,
                            int force_repr, void* out_link_contact_forces, void* out_mdot, int N, void* stream) {
  if (out_link_contact_forces == nullptr) return fail(JXS_EINVAL, "null out_link_contact_forces");
  if (model == nullptr) return fail(JXS_EINVAL, "null model");
  if (N <= 0) return fail(JXS_EINVAL, "N must be positive");
  if (out_mdot == nullptr)
    return run_any(model, jxs::MODE_DYN, state, nullptr, joint_torques, link_forces, force_repr, nullptr, nullptr,
                   out_link_contact_forces, nullptr, N, 1, stream, nullptr, true, 0, nullptr, 1.0);
  // with the deformation rates: the derivative block goes to a scratch block and its rows of m are copied out (one
  // strided device copy; the rows of the tangential deformation are the last 3 * n_points rows of every tile)
  jxs_layout lay;
  jxs_model_layout(model, &lay);
  const size_t esz = model->dtype == JXS_F64 ? 8 : 4;
  const size_t tiles = (size_t)((N + lay.tile - 1) / lay.tile);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (lay.n_points == 0) return run_any(model, jxs::MODE_DYN, state, nullptr, joint_torques, link_forces, force_repr, nullptr, nullptr,
                                        out_link_contact_forces, nullptr, N, 1, stream, nullptr, true, 0, nullptr, 1.0);
  void* xdot = nullptr;
  JXS_HIP(hipMalloc(&xdot, tiles * lay.tile * lay.n_rows * esz));
  int rc = run_any(model, jxs::MODE_DYN, state, xdot, joint_torques, link_forces, force_repr, nullptr, nullptr,
                   out_link_contact_forces, nullptr, N, 1, stream, nullptr, true, 0, nullptr, 1.0);
  if (rc == JXS_OK) {
    const size_t mrows = 3 * (size_t)lay.n_points;
    const hipError_t e = hipMemcpy2DAsync(out_mdot, esz * lay.tile * mrows, static_cast<const char*>(xdot) + esz * lay.tile * lay.row_m,
                                          esz * lay.tile * lay.n_rows, esz * lay.tile * mrows, tiles, hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) rc = hip_fail(e, "hipMemcpy2DAsync");
  }
  (void)hipStreamSynchronize(s);  // the scratch block is freed below
  (void)hipFree(xdot);
  return rc;
}
int jxs_inverse_dynamics(jxs_model* model, const void* state, const void* in_acc, const void* link_forces,
                         int force_repr, void* out_forces, int N, void* stream) {
  if (out_forces == nullptr) return fail(JXS_EINVAL, "null out_forces");
  return run_any(model, jxs::MODE_ID, state, nullptr, nullptr, link_forces, force_repr, in_acc, out_forces, nullptr,
                 nullptr, N, 1, stream);
}
int jxs_gravity_torques(jxs_model* model, const void* state, void* out_tau, int N, void* stream) {
  if (out_tau == nullptr) return fail(JXS_EINVAL, "null out_tau");
  // [round 3] a kernel of its own (MODE_GRAV) instead of the general RNEA evaluated at zero velocity: with v = 0 and
  // zero accelerations every link "accelerates" with -g, so the torques are subtree sums of the link weights --
  // no velocity rows, no prefix sums, no rotated inertias (jxs_core.h gravity_torques)
  return run_any(model, jxs::MODE_GRAV, state, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, N, 1, stream,
                 out_tau);
}
int jxs_debug_reload_env(void) {
  int k, b;
  debug_knobs(k, b);  // (the first read happens exactly once)
  load_knobs();
  return JXS_OK;
}
int jxs_solver_fault_counts(jxs_model* model, int* counts2, int reset, void* stream) {
  if (model == nullptr || counts2 == nullptr) return fail(JXS_EINVAL, "null argument");
  int* d = model->dtype == JXS_F64 ? model->f64->faults : model->f32->faults;
  hipStream_t s = static_cast<hipStream_t>(stream);
  JXS_HIP(hipMemcpyAsync(counts2, d, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
  if (reset) JXS_HIP(hipMemsetAsync(d, 0, 2 * sizeof(int), s));
  JXS_HIP(hipStreamSynchronize(s));
  return JXS_OK;
}
int jxs_mass_matrix(jxs_model* model, const void* state, void* out_M, int N, void* stream) {
  if (out_M == nullptr) return fail(JXS_EINVAL, "null out_M");
  if (model == nullptr) return fail(JXS_EINVAL, "null model");
  if (N <= 0) return fail(JXS_EINVAL, "N must be positive");
  jxs_layout lay;
  jxs_model_layout(model, &lay);
  const size_t nv = 6 + (size_t)lay.n_joints;
  const size_t elems = (size_t)((N + lay.tile - 1) / lay.tile) * lay.tile * nv * nv;
  // only the structurally non-zero entries are written by the kernel
  JXS_HIP(hipMemsetAsync(out_M, 0, elems * (model->dtype == JXS_F64 ? 8 : 4), static_cast<hipStream_t>(stream)));
  return run_any(model, jxs::MODE_CRBA, state, nullptr, nullptr, nullptr, 0, nullptr, out_M, nullptr, nullptr, N, 1, stream);
}
int jxs_mass_matrix_inverse(jxs_model* model, const void* state, void* out_Minv, int N, void* stream) {
  if (out_Minv == nullptr) return fail(JXS_EINVAL, "null out_Minv");
  if (model == nullptr) return fail(JXS_EINVAL, "null model");
  if (N <= 0) return fail(JXS_EINVAL, "N must be positive");
  jxs_layout lay;
  jxs_model_layout(model, &lay);
  const size_t nv = 6 + (size_t)lay.n_joints;
  const size_t elems = (size_t)((N + lay.tile - 1) / lay.tile) * lay.tile * nv * nv;
  JXS_HIP(hipMemsetAsync(out_Minv, 0, elems * (model->dtype == JXS_F64 ? 8 : 4), static_cast<hipStream_t>(stream)));
  return run_any(model, jxs::MODE_MINV, state, nullptr, nullptr, nullptr, 0, nullptr, out_Minv, nullptr, nullptr, N, 1, stream);
}
int jxs_jacobian_full(jxs_model* model, const void* state, void* out_J, void* out_B_H_L, int N, void* stream) {
  if (out_J == nullptr) return fail(JXS_EINVAL, "null out_J");
  if (model == nullptr) return fail(JXS_EINVAL, "null model");
  if (N <= 0) return fail(JXS_EINVAL, "N must be positive");
  jxs_layout lay;
  jxs_model_layout(model, &lay);
  const size_t nv = 6 + (size_t)lay.n_joints;
  const size_t elems = (size_t)((N + lay.tile - 1) / lay.tile) * lay.tile * 12 * nv;
  JXS_HIP(hipMemsetAsync(out_J, 0, elems * (model->dtype == JXS_F64 ? 8 : 4), static_cast<hipStream_t>(stream)));
  return run_any(model, jxs::MODE_JAC, state, nullptr, nullptr, nullptr, 0, nullptr, out_J, out_B_H_L, nullptr, N, 1, stream);
}
int jxs_refresh_kinematics(jxs_model* model, const void* state, void* out_link_transforms, void* out_link_velocities,
                           int N, void* stream) {
  return run_any(model, jxs::MODE_KIN, state, nullptr, nullptr, nullptr, 0, nullptr, nullptr, out_link_transforms,
                 out_link_velocities, N, 1, stream);
}

int jxs_comm_unique_id(char id[128]) {
  if (id == nullptr) return fail(JXS_EINVAL, "null id");
  int rc = rccl_load();
  if (rc != JXS_OK) return rc;
  ncclUniqueId u;
  ncclResult_t r = g_rccl.GetUniqueId(&u);
  if (r != ncclSuccess) return rccl_fail(r, "ncclGetUniqueId");
  static_assert(sizeof(u.internal) == 128, "unexpected ncclUniqueId size");
  std::memcpy(id, u.internal, 128);
  return JXS_OK;
}
int jxs_comm_version(int* version) {
  if (version == nullptr) return fail(JXS_EINVAL, "null version");
  int rc = rccl_load();
  if (rc != JXS_OK) return rc;
  if (g_rccl.GetVersion == nullptr) return fail(JXS_ECOMM, "librccl has no ncclGetVersion");
  ncclResult_t r = g_rccl.GetVersion(version);
  if (r != ncclSuccess) return rccl_fail(r, "ncclGetVersion");
  return JXS_OK;
}
int jxs_device_pci_bus_id(char* buf, int len) {
  if (buf == nullptr || len < 16) return fail(JXS_EINVAL, "buffer of at least 16 bytes expected");
  int dev = 0;
  JXS_HIP(hipGetDevice(&dev));
  JXS_HIP(hipDeviceGetPCIBusId(buf, len, dev));
  return JXS_OK;
}
int jxs_comm_init(void** comm, const char id[128], int rank, int world_size) {
  if (comm == nullptr || id == nullptr) return fail(JXS_EINVAL, "null argument");
  int rc = rccl_load();
  if (rc != JXS_OK) return rc;
  ncclUniqueId u;
  std::memcpy(u.internal, id, 128);
  ncclComm_t c;
  ncclResult_t r = g_rccl.CommInitRank(&c, world_size, u, rank);
  if (r != ncclSuccess) return rccl_fail(r, "ncclCommInitRank");
  *comm = c;
  return JXS_OK;
}
int jxs_comm_destroy(void* comm) {
  if (comm == nullptr) return JXS_OK;
  ncclResult_t r = g_rccl.CommDestroy(static_cast<ncclComm_t>(comm));
  if (r != ncclSuccess) return rccl_fail(r, "ncclCommDestroy");
  return JXS_OK;
}
int jxs_allgather(void* comm, const void* send, void* recv, uint64_t count, int dtype, void* stream) {
  if (comm == nullptr || send == nullptr || recv == nullptr) return fail(JXS_EINVAL, "null argument");
  ncclResult_t r = g_rccl.AllGather(send, recv, count, dtype == JXS_F64 ? ncclFloat64 : ncclFloat32,
                                    static_cast<ncclComm_t>(comm), static_cast<hipStream_t>(stream));
  if (r != ncclSuccess) return rccl_fail(r, "ncclAllGather");
  return JXS_OK;
}

}  // extern "C"
