// TEST INFRASTRUCTURE: host lockstep emulation of one lane group.
//
// `Vec<T,G>` holds one value per lane; every operator is elementwise, so the kernel core
// (jaxsim_amd/csrc/jxs_core.h), which is written branch-free over lane values, runs here on
// the CPU with exactly the data flow it has on the GPU: shuffles become gathers between
// lanes of the group.  This is what lets the CPU test-suite (-m "not gpu") check the kernel
// logic against the oracle without a GPU.  It is never shipped or benchmarked.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <vector>

#include "jxs_params.h"

namespace jxs {

template <typename T, int G>
struct Vec {
  T v[G];
  Vec() {
    for (int i = 0; i < G; ++i) v[i] = T();
  }
  Vec(T s) {  // NOLINT(google-explicit-constructor): scalar broadcast
    for (int i = 0; i < G; ++i) v[i] = s;
  }
#define JXS_BIN(op)                                           \
  friend Vec operator op(const Vec& a, const Vec& b) {        \
    Vec r;                                                    \
    for (int i = 0; i < G; ++i) r.v[i] = a.v[i] op b.v[i];    \
    return r;                                                 \
  }
  JXS_BIN(+) JXS_BIN(-) JXS_BIN(*) JXS_BIN(/) JXS_BIN(&) JXS_BIN(>>) JXS_BIN(^)
#undef JXS_BIN
#define JXS_CMP(op)                                              \
  friend Vec<bool, G> operator op(const Vec& a, const Vec& b) {  \
    Vec<bool, G> r;                                              \
    for (int i = 0; i < G; ++i) r.v[i] = a.v[i] op b.v[i];       \
    return r;                                                    \
  }
  JXS_CMP(<) JXS_CMP(<=) JXS_CMP(>) JXS_CMP(>=) JXS_CMP(==) JXS_CMP(!=)
#undef JXS_CMP
  friend Vec operator-(const Vec& a) {
    Vec r;
    for (int i = 0; i < G; ++i) r.v[i] = -a.v[i];
    return r;
  }
  Vec& operator+=(const Vec& b) { return *this = *this + b; }
  Vec& operator-=(const Vec& b) { return *this = *this - b; }
  Vec& operator*=(const Vec& b) { return *this = *this * b; }
  friend Vec vsel(const Vec<bool, G>& m, const Vec& a, const Vec& b) {
    Vec r;
    for (int i = 0; i < G; ++i) r.v[i] = m.v[i] ? a.v[i] : b.v[i];
    return r;
  }
  friend Vec vsqrt(const Vec& a) {
    Vec r;
    for (int i = 0; i < G; ++i) r.v[i] = std::sqrt(a.v[i]);
    return r;
  }
  friend Vec vrcp(const Vec& a) {
    Vec r;
    for (int i = 0; i < G; ++i) r.v[i] = T(1) / a.v[i];
    return r;
  }
  friend Vec vrcp_acc(const Vec& a) { return vrcp(a); }
  friend Vec vrsqrt(const Vec& a) {
    Vec r;
    for (int i = 0; i < G; ++i) r.v[i] = T(1) / std::sqrt(a.v[i]);
    return r;
  }
  friend Vec vabs(const Vec& a) {
    Vec r;
    for (int i = 0; i < G; ++i) r.v[i] = std::fabs(a.v[i]);
    return r;
  }
  friend Vec vmin(const Vec& a, const Vec& b) {
    Vec r;
    for (int i = 0; i < G; ++i) r.v[i] = std::fmin(a.v[i], b.v[i]);
    return r;
  }
  friend Vec vmax(const Vec& a, const Vec& b) {
    Vec r;
    for (int i = 0; i < G; ++i) r.v[i] = std::fmax(a.v[i], b.v[i]);
    return r;
  }
  friend Vec vpow(const Vec& a, const Vec& b) {
    Vec r;
    for (int i = 0; i < G; ++i) r.v[i] = std::pow(a.v[i], b.v[i]);
    return r;
  }
  friend Vec vsin(const Vec& a) {
    Vec r;
    for (int i = 0; i < G; ++i) r.v[i] = std::sin(a.v[i]);
    return r;
  }
  friend void vsincos(const Vec& a, Vec& s, Vec& c) {
    for (int i = 0; i < G; ++i) {
      s.v[i] = std::sin(a.v[i]);
      c.v[i] = std::cos(a.v[i]);
    }
  }
};

template <int G>
inline Vec<bool, G> operator&&(const Vec<bool, G>& a, const Vec<bool, G>& b) {
  Vec<bool, G> r;
  for (int i = 0; i < G; ++i) r.v[i] = a.v[i] && b.v[i];
  return r;
}
template <int G>
inline Vec<bool, G> operator||(const Vec<bool, G>& a, const Vec<bool, G>& b) {
  Vec<bool, G> r;
  for (int i = 0; i < G; ++i) r.v[i] = a.v[i] || b.v[i];
  return r;
}
template <int G>
inline Vec<bool, G> operator!(const Vec<bool, G>& a) {
  Vec<bool, G> r;
  for (int i = 0; i < G; ++i) r.v[i] = !a.v[i];
  return r;
}
template <int G>
inline Vec<bool, G> operator&&(const Vec<bool, G>& a, bool b) { return a && Vec<bool, G>(b); }
template <int G>
inline Vec<bool, G> operator&&(bool a, const Vec<bool, G>& b) { return Vec<bool, G>(a) && b; }
template <int G>
inline Vec<bool, G> operator||(const Vec<bool, G>& a, bool b) { return a || Vec<bool, G>(b); }

template <typename T_, int G_>
struct HostLanes {
  using T = T_;
  using V = Vec<T_, G_>;
  using VI = Vec<int, G_>;
  using VM = Vec<bool, G_>;
  static constexpr int G = G_;

  // small integers (lane and slot indices) kept in the LDS next to real numbers: exact both ways
  static V to_real(const VI& i) {
    V r;
    for (int k = 0; k < G_; ++k) r.v[k] = (T_)i.v[k];
    return r;
  }
  static VI to_int(const V& x) {
    VI r;
    for (int k = 0; k < G_; ++k) r.v[k] = (int)x.v[k];
    return r;
  }
  static bool mat3mul_packed(const V*, const V*, V*) { return false; }
  template <typename... Args>
  static bool axpy6_packed(Args...) { return false; }
  template <typename... Args>
  static bool axpy_range_packed(Args...) { return false; }
  template <typename... Args>
  static bool scale6_packed(Args...) { return false; }
  template <typename... Args>
  static bool add6_packed(Args...) { return false; }
  template <typename... Args>
  static bool dot6_packed(Args...) { return false; }

  // (the MFMA tiles of the contact solvers' Cholesky exist on the device only: the emulation runs the vector path,
  // which performs the same fused multiply-adds in the same order)
  static constexpr bool kHasMfma = false;
  static void lds_publish() {}
  static constexpr bool kHasRowShl = false;  // (the emulation gathers children through shfl: same values)
  template <int N>
  void fmac_row_shl(V*, const V*, const V&, int) const {}
  static constexpr bool kHasRowShift = false;  // (the emulation pulls through shfl: same values)
  void fmac7_from_next_slot(V*, const V*, const V&) const {}
  template <int NTMAX>
  struct ChTiles {
    ChTiles(const HostLanes&, int, int) {}
    void load() {}
    void extract(int) {}
    void update(int) {}
  };
  int env_;
  int N_;
  mutable std::vector<T_> lds_;
  // `lds_limit`: the words per environment the DEVICE launch of this mode allocates -- an access beyond it is recorded
  // (lds_oob_), which the harness turns into an error: the layout arithmetic of the kernels is checked on the CPU
  size_t lds_limit_;
  mutable long lds_oob_ = -1;
  HostLanes(int N, int env, size_t lds_words = 0, size_t lds_limit = (size_t)-1)
      : env_(env), N_(N), lds_(std::max((size_t)G_ * kRowRec + 64 + G_, lds_words), T_(0)), lds_limit_(lds_limit) {}
  // LDS writes of the lanes inside `m` only (device: one exec-masked region around the writes)
  mutable bool region_on_ = false;
  mutable VM region_;
  template <class F>
  void lds_masked(const VM& m, F&& f) const {
    region_on_ = true, region_ = m;
    f();
    region_on_ = false;
  }
  void dbg_store(T* out, int stride, int idx, const V& v, const VM& mask) const {
    for (int i = 0; i < G; ++i)
      if (mask.v[i]) out[(size_t)env_ * stride + idx] = v.v[i];
  }
  void lds_sync() const {}
  V env_sum(const V& x) const {
    T acc = T(0);
    for (int i = 0; i < G; ++i) acc += x.v[i];
    return V(acc);
  }
  V env_max(const V& x) const {
    T acc = x.v[0];
    for (int i = 1; i < G; ++i) acc = std::fmax(acc, x.v[i]);
    return V(acc);
  }
  V env_min(const V& x) const {
    T acc = x.v[0];
    for (int i = 1; i < G; ++i) acc = std::fmin(acc, x.v[i]);
    return V(acc);
  }
  V env_bcast16(const V& x, int src) const { return V(x.v[src]); }
  static unsigned uniform(unsigned x) { return x; }
  static unsigned long long uniform(unsigned long long x) { return x; }
  VI env_bits(const VM& m) const {
    int b = 0;
    for (int i = 0; i < G && i < 32; ++i) b |= m.v[i] ? (1 << i) : 0;
    return VI(b);
  }
  bool any(const VM& m) const {
    for (int i = 0; i < G; ++i)
      if (m.v[i]) return true;
    return false;
  }

  VI lane() const {
    VI r;
    for (int i = 0; i < G; ++i) r.v[i] = i;
    return r;
  }
  VM all_true() const { return VM(true); }
  void fence() const {}
  template <class KA>
  void stamp(const KA&, int) const {}
  template <class KA, class X>
  void stamp_after(const KA&, int, const X&) const {}
  template <class KA>
  void stamp_hwid(const KA&, int) const {}
  template <class KA>
  void debug_max(const KA&, int, int) const {}
  void count_fault(int* counters, int which, const VM& faulty) const {
    if (counters != nullptr && faulty.v[0]) counters[which] += 1;
  }

  template <typename U>
  Vec<U, G> shfl(const Vec<U, G>& x, const VI& src) const {
    Vec<U, G> r;
    for (int i = 0; i < G; ++i) r.v[i] = x.v[src.v[i] & (G - 1)];
    return r;
  }
  // lane shifts: out-of-group sources read as 0 here (on the device they read the neighbouring
  // group; every caller masks those lanes)
  V from_next(const V& x) const {
    V r;
    for (int i = 0; i < G; ++i) r.v[i] = (i + 1 < G) ? x.v[i + 1] : T(0);
    return r;
  }
  V from_prev(const V& x) const {
    V r;
    for (int i = 0; i < G; ++i) r.v[i] = (i >= 1) ? x.v[i - 1] : T(0);
    return r;
  }
  V row_bcast(const V& x, int k) const {  // lane k of the 16-lane row of every lane
    V r;
    for (int i = 0; i < G; ++i) r.v[i] = x.v[((i & ~15) + (k & 15)) & (G - 1)];
    return r;
  }
  template <int K, int N>
  void fmac_row_bcast(V* x, const V& m) const {  // x[i] += m * (x[i] of lane K of the 16-lane row)
    for (int i = 0; i < N; ++i) x[i] = x[i] + m * row_bcast(x[i], K);
  }
  V allreduce8(const V& x) const {
    V r;
    for (int i = 0; i < G; ++i) {
      T acc = T(0);
      for (int k = 0; k < 8; ++k) acc += x.v[(i & ~7) + k];
      r.v[i] = acc;
    }
    return r;
  }
  void allreduce8x3(V* x) const {
    for (int k = 0; k < 3; ++k) x[k] = allreduce8(x[k]);
  }
  void allreduce8x2(V* x) const {
    for (int k = 0; k < 2; ++k) x[k] = allreduce8(x[k]);
  }
  // x[j] += c1 * x[j]@s2 + c2 * x[j]@s1, j < 7 (s1, s2: source lanes; see the device backend)
  void ang_from_lin7(V* x, const V& c1, const V& c2, const VI& s1, const VI& s2) const {
    V g1[7], g2[7];
    for (int j = 0; j < 7; ++j) g1[j] = shfl(x[j], s1), g2[j] = shfl(x[j], s2);
    for (int j = 0; j < 7; ++j) x[j] = x[j] + c1 * g2[j] + c2 * g1[j];
  }
  void rank1_rows(V* m, const V& u, const V& s) const {  // m[j] += s * (u of lane j of the 8-lane slot)
    for (int j = 0; j < 6; ++j) {
      V g;
      for (int i = 0; i < G; ++i) g.v[i] = u.v[((i & ~7) + j) & (G - 1)];
      m[j] = m[j] + s * g;
    }
  }
  void allreduce8x7(V* x) const {
    for (int k = 0; k < 7; ++k) x[k] = allreduce8(x[k]);
  }
  void allreduce8x6(V* x) const {
    for (int k = 0; k < 6; ++k) x[k] = allreduce8(x[k]);
  }
  // two-wave workgroups: the emulation runs the inertia wave to completion, then the main wave, on the same
  // LDS image; a wait that is not satisfied by then is a protocol error of the kernel source
  mutable int flag_ = 0;
  mutable bool flag_error_ = false;
  void flag_post(int value) const { flag_ = value; }
  void flag_wait(int need, int& seen) const {
    seen = flag_;
    if (flag_ < need) flag_error_ = true;
  }
  static unsigned pin(unsigned x) { return x; }
  static int pin(int x) { return x; }
  void lds_touch(int a) const {
    if (a < 0 || (size_t)a >= lds_limit_) lds_oob_ = a;
  }
  void lds_write(const VI& addr, const V& v) const {
    for (int i = 0; i < G; ++i)
      if (!region_on_ || region_.v[i]) lds_touch(addr.v[i]), lds_[addr.v[i]] = v.v[i];
  }
  void lds_write(const VI& addr, const V& v, const VM& mask) const {
    for (int i = 0; i < G; ++i)
      if (mask.v[i] && (!region_on_ || region_.v[i])) lds_touch(addr.v[i]), lds_[addr.v[i]] = v.v[i];
  }
  V lds_read(const VI& addr) const {
    V r;
    for (int i = 0; i < G; ++i) lds_touch(addr.v[i]), r.v[i] = lds_[addr.v[i]];
    return r;
  }
  template <int N>
  void lds_writev(const VI& addr, const V* v) const {
    for (int k = 0; k < N; ++k) lds_write(addr + k, v[k]);
  }
  template <int N>
  void lds_writev_if(const VI& addr, const V* v, const VM& mask) const {
    for (int k = 0; k < N; ++k) lds_write(addr + k, v[k], mask);
  }
  // (device: lanes outside the mask write to the sink words; here they simply do not write -- nobody reads the sink)
  template <int N>
  void lds_writev_sel(const VI& addr, const V* v, const VM& mask, int) const {
    for (int k = 0; k < N; ++k) lds_write(addr + k, v[k], mask);
  }
  void lds_write_sel(const VI& addr, const V& v, const VM& mask, int) const { lds_write(addr, v, mask); }
  template <int N>
  void lds_readv(const VI& addr, V* v) const {
    for (int k = 0; k < N; ++k) v[k] = lds_read(addr + k);
  }
  // (`active`: lanes that take part; an inactive lane keeps its value and reads as zero from its neighbour, like a DPP
  // operand of an exec-masked lane with bound_ctrl)
  template <int N>
  void fmacN_from_next(V* a, const V* x, const V& m, const VM& active) const {
    for (int k = 0; k < N; ++k) {
      V xm = x[k];
      for (int i = 0; i < G; ++i)
        if (!active.v[i]) xm.v[i] = T(0);
      const V t = a[k] + m * from_next(xm);
      for (int i = 0; i < G; ++i)
        if (active.v[i]) a[k].v[i] = t.v[i];
    }
  }
  void fmac6_from_next(V* a, const V* x, const V& m, const VM& active) const { fmacN_from_next<6>(a, x, m, active); }
  void fmac9_from_next(V* a, const V* x, const V& m, const VM& active) const { fmacN_from_next<9>(a, x, m, active); }
  template <int OFF>
  void fmac6_row_from_next(V* w, const V& m) const {  // w[k] += m * w[k]@(lane + OFF) within the 16-lane row
    for (int k = 0; k < 6; ++k) w[k] = w[k] + m * row_from_next<OFF>(w[k]);
  }
  template <int OFF>
  V row_from_next(const V& x) const {
    V r;
    for (int i = 0; i < G; ++i) r.v[i] = ((i % 16) + OFF < 16 && i + OFF < G) ? x.v[i + OFF] : T(0);
    return r;
  }
  V lconstf(const T* tbl, int field) const {
    V r;
    for (int i = 0; i < G; ++i) r.v[i] = tbl[i * kLtfStride + field];
    return r;
  }
  VI lconsti(const int* tbl, int field) const {
    VI r;
    for (int i = 0; i < G; ++i) r.v[i] = lti_get(tbl + i * kLtiPackWords, field);
    return r;
  }
  VI rconsti(const int* tbl, int field) const {
    VI r;
    for (int i = 0; i < G; ++i) r.v[i] = rti_get(tbl + i * kRtiPackWords, field);
    return r;
  }
  VI hconsti(const int* head, int chunk) const {
    VI r;
    for (int i = 0; i < G; ++i) r.v[i] = head[chunk * G + i];
    return r;
  }
  V tgather(const T* tbl, const VI& idx) const {
    V r;
    for (int i = 0; i < G; ++i) r.v[i] = tbl[idx.v[i]];
    return r;
  }
  V ploadf(const T* tbl, int field, const VI& slot) const {
    V r;
    for (int i = 0; i < G; ++i) r.v[i] = tbl[slot.v[i] * kPtStride + field];
    return r;
  }
  VI ploadi(const int* tbl, int field, const VI& slot) const {
    VI r;
    for (int i = 0; i < G; ++i) r.v[i] = tbl[slot.v[i] * kPtStride + field];
    return r;
  }
  // tile-interleaved arrays: [N/TILE][rows][TILE], TILE = 64/G (see jxs_lanes_device.h)
  static constexpr int TILE = 64 / G;
  size_t at(int row, int nrows) const { return ((size_t)(env_ / TILE) * nrows + row) * TILE + (env_ % TILE); }
  V gload(const T* base, const VI& row, int nrows) const {
    V r;
    for (int i = 0; i < G; ++i) r.v[i] = base[at(row.v[i], nrows)];
    return r;
  }
  V gload_u(const T* base, int row, int nrows) const { return V(base[at(row, nrows)]); }
  void gstore(T* base, const VI& row, const V& val, const VM& mask, int nrows) const {
    for (int i = 0; i < G; ++i)
      if (mask.v[i]) base[at(row.v[i], nrows)] = val.v[i];
  }
};

}  // namespace jxs
