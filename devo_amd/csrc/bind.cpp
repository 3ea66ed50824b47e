// Compiled binding of libdevo_hip.so: the reference's three pybind11 extension modules — cuda_corr (devo/altcorr/correlation.cpp:57-63),
// cuda_ba (devo/fastba/ba.cpp:152-157), lietorch_backends (devo/lietorch/src/lietorch.cpp:286-316) — with the reference's function names,
// positional signatures and torch::Tensor arguments, as sub-modules of devo_amd._C, plus the same entry points as torch.ops.devo_hip.*.
// Every function allocates its outputs with ATen, takes c10::hip::getCurrentHIPStream() and calls the C ABI of include/devo_hip.h: no
// kernels here, no torch types below this file.  Errors surface as c10::Error -> Python RuntimeError, like the reference's TORCH_CHECK.
// devo_amd/backends/*.py (ctypes) stay as the no-compile form of this BINDING; neither has a CPU path.
//
// Host-side state kept here (what the reference's modules do not need because they read NCHW / fp32 directly):
//   * per tensor VERSION (storage address, version counter, sizes, strides, dtype; the cache keeps the source alive): the channel-blocked
//     (fp16) or split-blocked (fp32) copy of a pyramid level and the patch operand of the dense-product lookup kernel — DEVO rewrites both
//     once per frame, not per update iteration;
//   * the BA workspace of the last problem size per device.
#include <torch/extension.h>
#include <ATen/hip/HIPContext.h>
#include <c10/hip/HIPStream.h>
#include <c10/core/DeviceGuard.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <cstdlib>
#include <algorithm>
#include <list>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>
#include "../../include/devo_hip.h"

namespace {

using at::Tensor;
typedef std::vector<Tensor> TensorList;

// (ROCm builds of torch call their devices "cuda": the masquerading accessor is c10::hip::getCurrentHIPStream for that device type)
void* stream_of(const Tensor& t) { return (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream(); }

void check(int rc, const char* what) {
  TORCH_CHECK(rc == DEVO_OK, what, " failed (code ", rc, "): ", devo_last_error());
}

int dtype_code(const Tensor& t) {
  switch (t.scalar_type()) {
    case at::kFloat: return DEVO_F32;
    case at::kHalf: return DEVO_F16;
    case at::kDouble: return DEVO_F64;
    default: TORCH_CHECK(false, "devo_amd: unsupported dtype ", t.scalar_type());
  }
  return -1;
}

template <typename... Ts>
void require_gpu(const Ts&... ts) {
  for (const Tensor* t : {&ts...})
    TORCH_CHECK(!t->defined() || t->is_cuda(), "devo_amd: tensors must live on the GPU (the HIP path has no CPU fallback)");
}

Tensor idx(const Tensor& t) { return t.to(at::kLong).contiguous(); }
Tensor f32c(const Tensor& t) { return t.to(at::kFloat).contiguous(); }

bool env_off(const char* name) { const char* e = std::getenv(name); return e && e[0] == '0'; }
bool env_on(const char* name) { const char* e = std::getenv(name); return e && e[0] == '1'; }

// ------------------------------------------------------------------------------------------------ version-keyed caches
struct VersionKey {
  const void* ptr; int64_t version; std::vector<int64_t> sizes, strides; at::ScalarType dtype; int flavour;
  bool operator==(const VersionKey& o) const {
    return ptr == o.ptr && version == o.version && dtype == o.dtype && flavour == o.flavour && sizes == o.sizes && strides == o.strides;
  }
};
VersionKey key_of(const Tensor& t, int flavour) {
  return VersionKey{t.data_ptr(), t._version(), t.sizes().vec(), t.strides().vec(), t.scalar_type(), flavour};
}
struct Entry { VersionKey key; Tensor src, a, b; };     // (source kept alive: an equal key is the same storage with the same contents)
struct Lru {
  std::list<Entry> items; size_t cap; std::mutex mu;
  explicit Lru(size_t c) : cap(c) {}
  bool get(const VersionKey& k, Entry* out) {
    std::lock_guard<std::mutex> g(mu);
    for (auto it = items.begin(); it != items.end(); ++it)
      if (it->key == k) { items.splice(items.end(), items, it); *out = items.back(); return true; }
    return false;
  }
  void put(Entry e) {
    std::lock_guard<std::mutex> g(mu);
    for (auto it = items.begin(); it != items.end();) it = (it->key.ptr == e.key.ptr && it->key.flavour == e.key.flavour) ? items.erase(it) : std::next(it);   // older versions
    while (items.size() >= cap) items.pop_front();
    items.push_back(std::move(e));
  }
  void clear() { std::lock_guard<std::mutex> g(mu); items.clear(); }
};
Lru g_levels(4), g_patches(4), g_cl(4);

// ---- per-slot maintenance of a converted ring (round 6).  The reference writes ONE slot of its ring buffers per frame
// (`self.fmap1_[:, self.n % self.mem] = ...`, `self.gmap_[self.n % self.mem] = ...`: devo/devo.py:523-527; keyframe removal moves a few
// slots, :288-291) and then looks the whole ring up: re-converting 32 frames for one new one costs as much as the fp16 lookup itself.
// devo_amd.backends.install() wraps torch.Tensor.__setitem__ (devo_amd/backends/ring.py): a write into a tensor this cache holds a
// converted copy of is RECORDED here — (version counter after the write, element range) — and a later lookup whose key differs from a
// cached entry only in the version converts just the frames / patches the recorded writes touched, PROVIDED every version in between is
// accounted for by a record with a known range (each __setitem__ bumps the counter by exactly one; any other in-place operation leaves a
// gap and the whole tensor is converted again, as in rounds 1-5).  Sound by construction: nothing is assumed about unrecorded writes.
struct WriteRec { uint32_t version; int64_t off, len; };              // elements from the tensor's data_ptr; len < 0: region unknown
std::mutex g_wr_mu;
std::vector<std::pair<const void*, std::vector<WriteRec>>> g_writes;  // a handful of tensors: linear search
struct ConvStats { int64_t level_full = 0, level_frames = 0, patch_full = 0, patch_ranges = 0, patches_partial = 0; } g_conv;
constexpr size_t MAX_WRITE_RECS = 512;

bool is_tracked(const void* ptr) {
  for (Lru* c : {&g_levels, &g_patches}) {
    std::lock_guard<std::mutex> g(c->mu);
    for (const auto& e : c->items) if (e.key.ptr == ptr) return true;
  }
  return false;
}
void note_write(const void* ptr, uint32_t version_after, int64_t off, int64_t len) {
  std::lock_guard<std::mutex> g(g_wr_mu);
  for (auto& kv : g_writes)
    if (kv.first == ptr) {
      if (kv.second.size() >= MAX_WRITE_RECS) kv.second.clear();     // (a gap: the next lookup converts everything)
      kv.second.push_back({version_after, off, len});
      return;
    }
  g_writes.push_back({ptr, {{version_after, off, len}}});
}
void forget_writes(const void* ptr, uint32_t upto) {                  // records of versions <= upto are spent
  std::lock_guard<std::mutex> g(g_wr_mu);
  for (auto it = g_writes.begin(); it != g_writes.end(); ++it)
    if (it->first == ptr) {
      auto& v = it->second;
      v.erase(std::remove_if(v.begin(), v.end(), [&](const WriteRec& r) { return (int32_t)(r.version - upto) <= 0; }), v.end());
      if (v.empty()) g_writes.erase(it);
      return;
    }
}
// every version in (v0, v1] has a record with a known range -> their ranges; else false
bool writes_between(const void* ptr, uint32_t v0, uint32_t v1, std::vector<std::pair<int64_t, int64_t>>* ranges) {
  const uint32_t need = v1 - v0;
  if (need == 0 || need > MAX_WRITE_RECS) return false;
  std::lock_guard<std::mutex> g(g_wr_mu);
  for (auto& kv : g_writes)
    if (kv.first == ptr) {
      std::vector<char> seen(need, 0);
      for (const WriteRec& r : kv.second) {
        const uint32_t d = r.version - v0 - 1;                         // 0 .. need - 1 for the versions wanted
        if (d >= need) continue;
        if (r.len < 0 || seen[d]) return false;
        seen[d] = 1;
        ranges->push_back({r.off, r.off + r.len});
      }
      for (char c : seen) if (!c) return false;
      return true;
    }
  return false;
}
// an entry of the same tensor (storage, shape, strides, dtype, flavour) at an OLDER version
bool find_older(Lru& c, const VersionKey& k, Entry* out) {
  std::lock_guard<std::mutex> g(c.mu);
  for (auto& e : c.items)
    if (e.key.ptr == k.ptr && e.key.flavour == k.flavour && e.key.dtype == k.dtype && e.key.sizes == k.sizes && e.key.strides == k.strides && e.key.version != k.version) {
      *out = e;
      return true;
    }
  return false;
}
bool slot_updates() { static const bool on = !env_off("DEVO_RING_SLOTS"); return on; }

constexpr int64_t PLAN_MIN_EDGES = 2048, NCHW_CONVERT_MIN_EDGES = 1024;
bool mm_kernel() { static const bool on = !env_off("DEVO_CORR_MM") && !env_off("DEVO_CORR_MFMA"); return on; }

// One pyramid level as the C ABI wants it
struct Level { Tensor data, exps; int64_t strides[5]; int cblock; int H, W, n; };

Level describe(const Tensor& f, const Tensor& exps, int C, bool split) {
  Level l;
  l.data = f; l.exps = exps; l.cblock = 0;
  if (f.dim() == 6) {
    l.cblock = split ? DEVO_CBLOCK_SPLIT8 : (int)f.size(5);
    TORCH_CHECK(f.stride(5) == 1 && f.size(2) * f.size(5) == C, "cuda_corr: malformed channel-blocked fmap2");
  }
  for (int i = 0; i < 5; i++) l.strides[i] = f.stride(i);
  l.n = (int)f.size(1); l.H = (int)f.size(3); l.W = (int)f.size(4);
  return l;
}

// cuda_corr._fast_layout of the ctypes binding: NCHW fp16 -> cached channel-blocked copy; fp32 (any layout) -> cached split-blocked copy
// when the dense-product kernel will take the call
Level fast_level(const Tensor& fmap2, int64_t n_edges, bool allow_split) {
  const bool blocked = fmap2.dim() == 6;
  const int C = (int)(fmap2.size(2) * (blocked ? fmap2.size(5) : 1));
  const auto dt = fmap2.scalar_type();
  if ((dt != at::kHalf && dt != at::kFloat) || n_edges <= 0) return describe(fmap2, Tensor(), C, false);
  const bool want_split = allow_split && mm_kernel() && dt == at::kFloat && C % 32 == 0 && C <= 128 && fmap2.numel() > 0;
  const int64_t B = fmap2.size(0), n = fmap2.size(1), H = fmap2.size(3), W = fmap2.size(4);
  if (!want_split) {
    const bool nchw = !blocked && C % 8 == 0 && fmap2.stride(2) == H * W && fmap2.stride(3) == W && fmap2.stride(4) == 1 && B * n > 0;
    if (!nchw || n_edges < NCHW_CONVERT_MIN_EDGES || env_on("DEVO_CORR_NCHW_DIRECT")) return describe(fmap2, Tensor(), C, false);
  }
  const VersionKey k = key_of(fmap2, want_split ? 1 : 0);
  Entry e;
  if (g_levels.get(k, &e)) return describe(e.a, e.b, C, want_split);
  void* st = stream_of(fmap2);
  // the same ring at an older version whose every write since is on record: convert the frames those writes touched, in place
  if (slot_updates() && !blocked && fmap2.is_contiguous() && find_older(g_levels, k, &e) && devo_stream_capturing(st) == 0) {
    std::vector<std::pair<int64_t, int64_t>> ranges;
    if (writes_between(k.ptr, (uint32_t)e.key.version, (uint32_t)k.version, &ranges)) {
      const int64_t per = (int64_t)C * H * W, total = B * n;
      std::vector<char> dirty((size_t)total, 0);
      for (auto& r : ranges)
        for (int64_t f = std::max<int64_t>(0, r.first / per); f < total && f * per < r.second; f++) dirty[(size_t)f] = 1;
      const int64_t es = fmap2.element_size();
      for (int64_t f = 0; f < total; f++) {
        if (!dirty[(size_t)f]) continue;
        int64_t f1 = f;
        while (f1 + 1 < total && dirty[(size_t)(f1 + 1)] && (f1 + 1) / n == f / n) f1++;              // a run of frames of one batch entry
        const int64_t b = f / n, f0 = f - b * n, cnt = f1 - f + 1;
        const char* src = (const char*)fmap2.data_ptr() + (b * fmap2.stride(0) + f0 * fmap2.stride(1)) * es;
        char* dst = (char*)e.a.data_ptr() + (b * e.a.stride(0) + f0 * e.a.stride(1)) * es;
        if (want_split) {
          const int64_t f2s[4] = {fmap2.stride(1), fmap2.stride(2), fmap2.stride(3), fmap2.stride(4)};
          Tensor scratch = at::empty({cnt}, e.b.options());
          check(devo_corr_pyramid_split_frames(src, f2s, 0, (int)cnt, C, (int)H, (int)W, dst, e.a.stride(1), e.b.data_ptr<int>() + b * n + f0, scratch.data_ptr<int>(), st),
                "cuda_corr: fp32 ring slot -> split-blocked");
        } else {
          check(devo_pyramid_build(src, dst, nullptr, (int)cnt, C, (int)H, (int)W, fmap2.stride(1), e.a.stride(1), 0, dtype_code(fmap2), st), "cuda_corr: NCHW ring slot -> channel-blocked");
        }
        g_conv.level_frames += cnt;
        f = f1;
      }
      e.key = k; e.src = fmap2;
      g_levels.put(e);                                          // (replaces the older version's entry: same converted tensors)
      forget_writes(k.ptr, (uint32_t)k.version);
      return describe(e.a, e.b, C, want_split);
    }
  }
  e = Entry();
  e.key = k; e.src = fmap2;
  g_conv.level_full++;
  forget_writes(k.ptr, (uint32_t)k.version);
  if (want_split) {
    e.a = at::empty({B, n, C / 8, H, W, 8}, fmap2.options());
    e.b = at::empty({B * n + n}, fmap2.options().dtype(at::kInt));
    const int64_t f2s[4] = {fmap2.stride(1), fmap2.stride(2), fmap2.stride(3), fmap2.stride(4)};
    for (int64_t b = 0; b < B; b++)                           // ascending: call b's scratch is the (not yet written) slice of b + 1
      check(devo_corr_pyramid_split((const char*)fmap2.data_ptr() + b * fmap2.stride(0) * 4, f2s, blocked ? (int)fmap2.size(5) : 0, (int)n, C, (int)H, (int)W,
                                    (char*)e.a.data_ptr() + b * e.a.stride(0) * 4, e.a.stride(1), e.b.data_ptr<int>() + b * n, st), "cuda_corr: fp32 level -> split-blocked");
  } else {
    e.a = at::empty({B, n, C / 8, H, W, 8}, fmap2.options());
    const int64_t es = fmap2.element_size();
    for (int64_t b = 0; b < B; b++)
      check(devo_pyramid_build((const char*)fmap2.data_ptr() + b * fmap2.stride(0) * es, (char*)e.a.data_ptr() + b * e.a.stride(0) * es, nullptr, (int)n, C, (int)H, (int)W,
                               fmap2.stride(1), e.a.stride(1), 0, dtype_code(fmap2), st), "cuda_corr: NCHW -> channel-blocked");
  }
  g_levels.put(e);
  return describe(e.a, e.b, C, want_split);
}

Tensor patch_operand(const Tensor& fmap1) {            // undefined: the lookup takes the other kernels
  const int64_t C = fmap1.size(2);
  if (!mm_kernel() || fmap1.size(3) != 3 || fmap1.size(4) != 3 || C % 32 != 0 || fmap1.numel() == 0 ||
      (fmap1.scalar_type() != at::kHalf && fmap1.scalar_type() != at::kFloat))
    return Tensor();
  const VersionKey k = key_of(fmap1, 2);
  Entry e;
  if (g_patches.get(k, &e)) return e.a;
  const int64_t n = fmap1.size(0) * fmap1.size(1);
  void* st = stream_of(fmap1);
  if (slot_updates() && fmap1.is_contiguous() && find_older(g_patches, k, &e) && devo_stream_capturing(st) == 0) {   // (see fast_level)
    std::vector<std::pair<int64_t, int64_t>> ranges;
    if (writes_between(k.ptr, (uint32_t)e.key.version, (uint32_t)k.version, &ranges)) {
      const int64_t per = C * 9;
      for (auto& r : ranges) {
        const int64_t p0 = std::max<int64_t>(0, r.first / per), p1 = std::min<int64_t>(n, (r.second + per - 1) / per);
        if (p1 <= p0) continue;
        check(devo_corr_patch_transpose_range(fmap1.data_ptr(), e.a.data_ptr(), (int)n, (int)p0, (int)(p1 - p0), (int)C, dtype_code(fmap1), st), "cuda_corr.patches_transposed (slot)");
        g_conv.patch_ranges++; g_conv.patches_partial += p1 - p0;
      }
      e.key = k; e.src = fmap1;
      g_patches.put(e);
      forget_writes(k.ptr, (uint32_t)k.version);
      return e.a;
    }
  }
  e = Entry();
  const size_t nbytes = devo_corr_patch_operand_bytes((int)n, (int)C, dtype_code(fmap1));
  TORCH_CHECK(nbytes > 0, "cuda_corr: C = ", C, " unsupported by the patch operand (C % 8)");
  e.key = k; e.src = fmap1;
  g_conv.patch_full++;
  forget_writes(k.ptr, (uint32_t)k.version);
  e.a = at::empty({(int64_t)nbytes}, fmap1.options().dtype(at::kByte));
  check(devo_corr_patch_transpose(fmap1.data_ptr(), e.a.data_ptr(), (int)n, (int)C, dtype_code(fmap1), stream_of(fmap1)), "cuda_corr.patches_transposed");
  g_patches.put(e);
  return e.a;
}

Tensor make_plan(const Tensor& coords, const Tensor& jj, int n_frames, int height, float coord_scale, int radius) {
  const int64_t B = coords.size(0), E = coords.size(1);
  Tensor order = at::empty({2 * B * E + 2}, coords.options().dtype(at::kInt));
  check(devo_corr_order(coords.data_ptr<float>(), jj.data_ptr<int64_t>(), order.data_ptr<int>(), (int)B, (int)E, n_frames, (int)coords.size(3), height, coord_scale,
                        radius, 0, 0, stream_of(coords)), "cuda_corr.plan");
  return order;
}

void corr_prep(const Tensor& fmap1, const Tensor& fmap2, const Tensor& coords, bool allow_blocked) {
  TORCH_CHECK(fmap1.scalar_type() == fmap2.scalar_type(), "cuda_corr: fmap1 and fmap2 must have the same dtype");
  TORCH_CHECK(fmap1.dim() == 5 && (fmap2.dim() == 5 || (allow_blocked && fmap2.dim() == 6)) && coords.dim() == 5,
              "cuda_corr: expected fmap1 [B,Np,C,P,P], fmap2 [B,n,C,H,W], coords [B,E,2,P,P]");
}

struct LastPlan { const void* jj = nullptr; uint32_t ver = 0; int64_t BE = 0; int n = 0, radius = 0; void* stream = nullptr; Tensor plan; };
thread_local LastPlan g_last_plan;                                    // the plan a per-level call made, for the call right behind it

// corr forward writing element l of edge (b, e) at out[(b E + e) estride + l lstride + offset]
void corr_forward_into(Tensor& out, const Tensor& fmap1_, const Tensor& fmap2, const Tensor& coords_, const Tensor& ii_, const Tensor& jj_, int radius,
                       int64_t estride, int64_t lstride, int64_t offset, const Tensor& order_, double coord_div) {
  require_gpu(fmap1_, fmap2, coords_, ii_, jj_);
  corr_prep(fmap1_, fmap2, coords_, true);
  c10::DeviceGuard guard(fmap1_.device());
  const Tensor fmap1 = fmap1_.contiguous(), coords = f32c(coords_), ii = idx(ii_), jj = idx(jj_);
  const int64_t B = coords.size(0), E = coords.size(1), P = coords.size(3), Np = fmap1.size(1), C = fmap1.size(2);
  const Tensor f1t = patch_operand(fmap1);
  const Level lv = fast_level(fmap2, B * E, f1t.defined() && lstride > 0);
  Tensor order = order_;
  if (!order.defined() && B * E >= PLAN_MIN_EDGES) {
    // DEVO calls corr once per pyramid level with the SAME index tensors (devo.py:215-216): the second call takes the plan the first one made
    // — "one plan serves every level of a pyramid", and a plan only decides which edges run together: the result does not depend on it by
    // one bit (tests/test_gpu_altcorr.py::test_lookup_results_do_not_depend_on_the_plan).  A plan is handed on ONCE, to the call right behind
    // the one that made it, and only for the same jj tensor (storage, version, size), frame count and stream.
    auto& last = g_last_plan;
    void* st = stream_of(coords);
    if (devo_stream_capturing(st) != 0) {                     // a capture executes nothing: a plan made now holds garbage until the graph runs
      last = LastPlan();                                      // — neither handed on nor taken over
      order = make_plan(coords, jj, lv.n, lv.H, (float)coord_div, radius);
    } else if (last.plan.defined() && last.jj == jj_.data_ptr() && last.ver == jj_._version() && last.BE == B * E && last.n == lv.n && last.radius == radius && last.stream == st &&
        last.plan.device() == coords.device()) {
      order = last.plan;
      last.plan = Tensor();
    } else {
      order = make_plan(coords, jj, lv.n, lv.H, (float)coord_div, radius);
      last.jj = jj_.data_ptr(); last.ver = jj_._version(); last.BE = B * E; last.n = lv.n; last.radius = radius; last.stream = st; last.plan = order;
    }
  }
  check(devo_corr_forward(fmap1.data_ptr(), lv.data.data_ptr(), coords.data_ptr<float>(), ii.data_ptr<int64_t>(), jj.data_ptr<int64_t>(), out.data_ptr(), (int)B, (int)E,
                          (int)Np, lv.n, (int)C, (int)P, lv.H, lv.W, lv.strides, lv.cblock, estride, lstride, offset, radius, dtype_code(fmap1),
                          order.defined() ? order.data_ptr<int>() : nullptr, (float)coord_div, f1t.defined() ? f1t.data_ptr() : nullptr,
                          lv.exps.defined() ? lv.exps.data_ptr<int>() : nullptr, stream_of(fmap1)), "cuda_corr.forward");
}

// ------------------------------------------------------------------------------------------------ cuda_corr (correlation.cpp:57-63)
TensorList corr_forward(Tensor fmap1, Tensor fmap2, Tensor coords, Tensor ii, Tensor jj, int64_t radius) {       // correlation.cpp:58
  const int64_t B = coords.size(0), E = coords.size(1), P = coords.size(3), Dm = 2 * radius + 1;
  Tensor out = at::empty({B, E, Dm, Dm, P, P}, fmap1.options());
  corr_forward_into(out, fmap1, fmap2, coords, ii, jj, (int)radius, Dm * Dm * P * P, 1, 0, Tensor(), 1.0);
  return {out};
}

thread_local int g_last_bwd_path = -1;

TensorList corr_backward(Tensor fmap1_, Tensor fmap2_, Tensor coords_, Tensor ii_, Tensor jj_, Tensor grad_, int64_t radius) {   // correlation.cpp:59
  require_gpu(fmap1_, fmap2_, coords_, ii_, jj_, grad_);
  corr_prep(fmap1_, fmap2_, coords_, false);
  if (fmap1_.scalar_type() != at::kFloat) {
    // the reference dispatches its backward over half / float / double with a FLOAT gradient accessor (correlation_kernel.cu:146,280);
    // here the kernel is fp32: other dtypes are computed in fp32 and cast back
    const auto dt = fmap1_.scalar_type();
    TensorList g = corr_backward(fmap1_.to(at::kFloat), fmap2_.to(at::kFloat), coords_, ii_, jj_, grad_.to(at::kFloat), radius);
    return {g[0].to(dt), g[1].to(dt)};
  }
  c10::DeviceGuard guard(fmap1_.device());
  const Tensor fmap1 = fmap1_.contiguous(), coords = f32c(coords_), ii = idx(ii_), jj = idx(jj_), grad = f32c(grad_);
  Tensor fmap2 = fmap2_;
  const int64_t B = coords.size(0), E = coords.size(1), P = coords.size(3), Np = fmap1.size(1), C = fmap1.size(2);
  const int64_t n2 = fmap2.size(1), H2 = fmap2.size(3), W2 = fmap2.size(4);
  auto is_cl = [&](const Tensor& f) { return f.stride(2) == 1 && f.stride(4) == C && f.stride(3) == W2 * C && f.stride(1) >= H2 * W2 * C; };
  bool cl = is_cl(fmap2);
  if (!cl && C % 128 == 0 && B * E >= NCHW_CONVERT_MIN_EDGES && !env_on("DEVO_CORR_NCHW_DIRECT") &&
      devo_corr_backward_workspace_bytes((int)B, (int)E, (int)Np, (int)n2, (int)C, (int)radius, 1) > 0) {
    // the product form reads a channels-last copy, cached per version of the tensor: a training step calls backward once per update
    // iteration and level on the SAME pyramid tensors (enet.py:203-216)
    const VersionKey k = key_of(fmap2, 3);
    Entry e;
    if (!g_cl.get(k, &e)) {
      e.key = k; e.src = fmap2.detach();
      e.a = fmap2.detach().permute({0, 1, 3, 4, 2}).contiguous().permute({0, 1, 4, 2, 3});
      g_cl.put(e);
    }
    fmap2 = e.a;
    cl = true;
  }
  Tensor d1 = at::empty_like(fmap1), d2;
  if (cl) {
    d2 = at::empty({B, n2, H2, W2, C}, fmap2.options()).permute({0, 1, 4, 2, 3});
    if (d2.strides() != fmap2.strides()) d2 = at::empty_strided(fmap2.sizes(), fmap2.strides(), fmap2.options());
  } else {
    d2 = at::empty_strided(fmap2.sizes(), fmap2.strides(), fmap2.options());
  }
  int64_t span = 1;
  for (int i = 0; i < 5; i++) span += (fmap2.size(i) - 1) * fmap2.stride(i);
  const size_t nws = devo_corr_backward_workspace_bytes((int)B, (int)E, (int)Np, (int)n2, (int)C, (int)radius, cl ? 1 : 0);
  Tensor ws = nws ? at::empty({(int64_t)nws}, fmap1.options().dtype(at::kByte)) : Tensor();
  const int64_t f2s[5] = {fmap2.stride(0), fmap2.stride(1), fmap2.stride(2), fmap2.stride(3), fmap2.stride(4)};
  check(devo_corr_backward(fmap1.data_ptr(), fmap2.data_ptr(), coords.data_ptr<float>(), ii.data_ptr<int64_t>(), jj.data_ptr<int64_t>(), grad.data_ptr<float>(),
                           d1.data_ptr(), d2.data_ptr(), (int)B, (int)E, (int)Np, (int)n2, (int)C, (int)P, (int)H2, (int)W2, f2s, span, (int)radius, dtype_code(fmap1),
                           nws ? ws.data_ptr() : nullptr, nws, stream_of(fmap1)), "cuda_corr.backward");
  g_last_bwd_path = devo_corr_backward_last_path();
  return {d1, d2};
}

std::string corr_last_backward_path() {
  static const char* names[3] = {"atomic", "segments", "product"};
  return (g_last_bwd_path >= 0 && g_last_bwd_path < 3) ? names[g_last_bwd_path] : "";
}

TensorList patchify_forward(Tensor net, Tensor coords_, int64_t radius) {                  // correlation.cpp:61
  require_gpu(net, coords_);
  TORCH_CHECK(net.dim() == 4 && coords_.dim() == 3, "cuda_corr.patchify_forward: expected net [B,C,H,W], coords [B,M,2]");
  c10::DeviceGuard guard(net.device());
  const Tensor coords = f32c(coords_);
  const int64_t B = coords.size(0), M = coords.size(1), C = net.size(1), H = net.size(2), W = net.size(3), D = 2 * radius + 2;
  Tensor out = at::empty({B, M, C, D, D}, net.options());
  const int64_t ns[4] = {net.stride(0), net.stride(1), net.stride(2), net.stride(3)};
  check(devo_patchify_forward(net.data_ptr(), coords.data_ptr<float>(), out.data_ptr(), (int)B, (int)M, (int)C, (int)H, (int)W, ns, (int)radius, dtype_code(net),
                              stream_of(net)), "cuda_corr.patchify_forward");
  return {out};
}

TensorList patchify_backward(Tensor net, Tensor coords_, Tensor gradient_, int64_t radius) {    // correlation.cpp:62
  require_gpu(net, coords_, gradient_);
  if (net.scalar_type() == at::kHalf)                     // scattered in fp32 (hardware float atomics), cast back like the forward's dtype
    return {patchify_backward(net.to(at::kFloat), coords_, gradient_.to(at::kFloat), radius)[0].to(at::kHalf)};
  c10::DeviceGuard guard(net.device());
  const Tensor coords = f32c(coords_), gradient = gradient_.to(net.scalar_type()).contiguous();
  const int64_t B = coords.size(0), M = coords.size(1), C = net.size(1), H = net.size(2), W = net.size(3);
  // the gradient in net's own layout when that is a dense permutation (channels-last from the encoders' convolutions), contiguous otherwise
  int64_t span = 1;
  for (int i = 0; i < 4; i++) span += (net.size(i) - 1) * net.stride(i);
  const bool dense = net.numel() > 0 && span == net.numel();
  Tensor out = dense ? at::empty_strided(net.sizes(), net.strides(), net.options()) : at::empty({B, C, H, W}, net.options());
  const int64_t gs[4] = {out.stride(0), out.stride(1), out.stride(2), out.stride(3)};
  check(devo_patchify_backward(coords.data_ptr<float>(), gradient.data_ptr(), out.data_ptr(), (int)B, (int)M, (int)C, (int)H, (int)W, gs, (int)radius, dtype_code(net),
                               stream_of(net)), "cuda_corr.patchify_backward");
  return {out};
}

// ------------------------------------------------------------------------------------------------ cuda_ba (ba.cpp:152-157)
// One live workspace per (problem size, device, STREAM): two streams never share scratch, and a workspace that is replaced stays alive until
// the work enqueued on it has run (the caching allocator frees a block for reuse on the stream it was allocated on: a stream-ordered free).
struct WsKey { int E, Np, N, dev; void* stream; bool operator==(const WsKey& o) const { return E == o.E && Np == o.Np && N == o.N && dev == o.dev && stream == o.stream; } };
std::mutex g_ws_mu;
WsKey g_ws_key{-1, -1, -1, -1, nullptr};
Tensor g_ws;

Tensor ba_workspace(int E, int Np, int N, const at::Device& dev, void* stream) {
  std::lock_guard<std::mutex> g(g_ws_mu);
  const WsKey k{E, Np, N, (int)dev.index(), stream};
  if (g_ws.defined() && g_ws_key == k) return g_ws;
  const size_t nbytes = devo_ba_workspace_bytes(E, Np, N);
  TORCH_CHECK(nbytes > 0, "cuda_ba: unsupported problem size (E=", E, ", Np=", Np, ", N=", N, "; at most 128 optimised poses)");
  g_ws = at::empty({(int64_t)nbytes}, at::TensorOptions().dtype(at::kByte).device(dev));     // one live workspace: the graph size changes once per frame
  g_ws_key = k;
  return g_ws;
}

// The index half of cuda_ba.forward (unique patches, edges grouped by patch: ba_cuda.cu:435-437) depends on kk alone, and DEVO calls the BA
// again and again on one graph (18 update iterations per training sequence, train.py / enet.py:313-361; 12 at initialisation, devo.py:545).
// The prepared tables of the LAST call are kept with the workspace they live in: key = (kk's storage address and version counter, E, patch
// slots, window size, workspace, stream); the cache keeps kk and the workspace alive, so an equal key is the same storage with the same
// contents.  A hit runs devo_ba_forward_prepared: one launch (12.6 us at cfg2) less per call.  Never consulted or filled while the stream is
// being captured into a graph (a capture executes nothing: the tables would not exist).  DEVO_BA_PREP_CACHE=0 switches it off.
struct PrepKey {
  const void* kk = nullptr; uint32_t ver = 0; int64_t E = -1; int Np = -1, N = -1; const void* ws = nullptr; void* stream = nullptr;
  bool operator==(const PrepKey& o) const { return kk == o.kk && ver == o.ver && E == o.E && Np == o.Np && N == o.N && ws == o.ws && stream == o.stream; }
};
std::mutex g_prep_mu;
PrepKey g_prep_key;
Tensor g_prep_kk, g_prep_ws;
int64_t g_prep_hits = 0, g_prep_misses = 0;
void ba_prep_invalidate() { std::lock_guard<std::mutex> g(g_prep_mu); g_prep_key = PrepKey(); g_prep_kk = Tensor(); g_prep_ws = Tensor(); }

// ba.cpp:153.  Mutates poses ([1,Nbuf,7]) and patches ([1,Np,3,P,P]) in place and returns [] (devo/fastba/ba.py:7-8 passes poses.data;
// devo/devo.py:337 relies on the mutation).  ws / status / prepared: extras of this package's callers (devo_amd.fastba).
TensorList ba_forward(Tensor poses, Tensor patches, Tensor intrinsics_, Tensor target_, Tensor weight_, Tensor lmbda_, Tensor ii_, Tensor jj_, Tensor kk_,
                      int64_t t0, int64_t t1, int64_t iterations, c10::optional<Tensor> ws_, c10::optional<Tensor> status_, bool prepared) {
  require_gpu(poses, patches, intrinsics_, target_, weight_, lmbda_, ii_, jj_, kk_);
  TORCH_CHECK(poses.scalar_type() == at::kFloat && poses.is_contiguous(), "cuda_ba.forward: poses must be a contiguous float32 tensor (it is updated in place)");
  TORCH_CHECK(patches.scalar_type() == at::kFloat && patches.is_contiguous(), "cuda_ba.forward: patches must be a contiguous float32 tensor (it is updated in place)");
  c10::DeviceGuard guard(poses.device());
  const int64_t P = patches.size(-1), Nbuf = poses.numel() / 7, Np = patches.numel() / (3 * P * P);
  const Tensor ii = idx(ii_), jj = idx(jj_), kk = idx(kk_);
  const int64_t E = ii.numel();
  const Tensor intrinsics = f32c(intrinsics_), target = f32c(target_), weight = f32c(weight_), lmbda = f32c(lmbda_.reshape({-1}));
  TORCH_CHECK(!(prepared && !(ws_.has_value() && ws_->defined())), "cuda_ba.forward: prepared=True needs the workspace that prepare() filled");
  void* st = stream_of(poses);
  Tensor ws = (ws_.has_value() && ws_->defined()) ? *ws_ : ba_workspace((int)E, (int)Np, (int)(t1 - t0), poses.device(), st);
  int* status = (status_.has_value() && status_->defined()) ? status_->data_ptr<int>() : nullptr;
  static const bool prep_cache = !env_off("DEVO_BA_PREP_CACHE");
  if (!prepared && prep_cache && E > 0 && iterations > 0) {
    if (devo_stream_capturing(st) != 0) ba_prep_invalidate();          // (a graph replay rewrites the workspace's tables behind the cache's back)
    else if (!kk_.is_inference()) {
      const PrepKey k{kk_.data_ptr(), (uint32_t)kk_._version(), E, (int)Np, (int)(t1 - t0), ws.data_ptr(), st};
      std::lock_guard<std::mutex> g(g_prep_mu);
      if (g_prep_key == k) { prepared = true; g_prep_hits++; }
      else { g_prep_key = k; g_prep_kk = kk_; g_prep_ws = ws; g_prep_misses++; }     // this call prepares: the tables are in `ws` behind it
    }
  } else if (prepared) ba_prep_invalidate();                           // the caller's own prepare() filled the workspace: not the cached graph's tables
  auto fn = prepared ? devo_ba_forward_prepared : devo_ba_forward;
  check(fn(poses.data_ptr<float>(), patches.data_ptr<float>(), intrinsics.data_ptr<float>(), target.data_ptr<float>(), weight.data_ptr<float>(), lmbda.data_ptr<float>(),
           ii.data_ptr<int64_t>(), jj.data_ptr<int64_t>(), kk.data_ptr<int64_t>(), (int)E, (int)Nbuf, (int)Np, (int)P, (int)t0, (int)t1, (int)iterations, ws.data_ptr(),
           (size_t)ws.numel(), status, st), "cuda_ba.forward");
  return {};
}

TensorList ba_neighbors(Tensor ii_, Tensor jj_) {                       // ba.cpp:154 -> [ix, jx] (int64, on the GPU); no device<->host round trip
  require_gpu(ii_, jj_);
  c10::DeviceGuard guard(ii_.device());
  const Tensor ii = idx(ii_), jj = idx(jj_);
  const int64_t E = ii.numel();
  Tensor ix = at::empty({E}, ii.options()), jx = at::empty({E}, ii.options());
  Tensor ws = at::empty({(int64_t)devo_neighbors_workspace_bytes((int)E)}, ii.options().dtype(at::kByte));
  check(devo_ba_neighbors(ii.data_ptr<int64_t>(), jj.data_ptr<int64_t>(), ix.data_ptr<int64_t>(), jx.data_ptr<int64_t>(), (int)E, ws.data_ptr(), (size_t)ws.numel(),
                          stream_of(ii)), "cuda_ba.neighbors");
  return {ix, jx};
}

Tensor ba_reproject(Tensor poses_, Tensor patches_, Tensor intrinsics_, Tensor ii_, Tensor jj_, Tensor kk_) {      // ba.cpp:155 -> coords [1, E, 2, P, P]
  require_gpu(poses_, patches_, intrinsics_, ii_, jj_, kk_);
  c10::DeviceGuard guard(poses_.device());
  const int64_t P = patches_.size(-1);
  const Tensor ii = idx(ii_), jj = idx(jj_), kk = idx(kk_), poses = f32c(poses_), patches = f32c(patches_), intrinsics = f32c(intrinsics_);
  const int64_t E = ii.numel();
  Tensor coords = at::empty({1, E, 2, P, P}, poses.options());
  check(devo_ba_reproject(poses.data_ptr<float>(), patches.data_ptr<float>(), intrinsics.data_ptr<float>(), ii.data_ptr<int64_t>(), jj.data_ptr<int64_t>(),
                          kk.data_ptr<int64_t>(), coords.data_ptr<float>(), (int)E, (int)P, stream_of(poses)), "cuda_ba.reproject");
  return coords;
}

// the fused form of devo/projective_ops.py:53-105 that DEVO.update really calls (no plan, no Jacobians: devo_amd.backends.cuda_ba.transform
// has the full argument list); layout "pp2": [1,E,P,P,2], "2pp": [1,E,2,P,P]
Tensor ba_transform(Tensor poses_, Tensor patches_, Tensor intrinsics_, Tensor ii_, Tensor jj_, Tensor kk_, bool layout_2pp) {
  require_gpu(poses_, patches_, intrinsics_, ii_, jj_, kk_);
  c10::DeviceGuard guard(poses_.device());
  const int64_t P = patches_.size(-1);
  const Tensor ii = idx(ii_), jj = idx(jj_), kk = idx(kk_), poses = f32c(poses_), patches = f32c(patches_), intrinsics = f32c(intrinsics_);
  const int64_t E = ii.numel();
  Tensor c = layout_2pp ? at::empty({1, E, 2, P, P}, poses.options()) : at::empty({1, E, P, P, 2}, poses.options());
  check(devo_transform(poses.data_ptr<float>(), patches.data_ptr<float>(), intrinsics.data_ptr<float>(), ii.data_ptr<int64_t>(), jj.data_ptr<int64_t>(), kk.data_ptr<int64_t>(),
                       layout_2pp ? nullptr : c.data_ptr<float>(), layout_2pp ? c.data_ptr<float>() : nullptr, nullptr, nullptr, nullptr, nullptr, (int)E, (int)P, 0,
                       nullptr, 0, 0, 0, 0, 0, stream_of(poses)), "cuda_ba.transform");
  return c;
}

// ------------------------------------------------------------------------------------------------ lietorch_backends (lietorch.cpp:286-316), SE3 only
constexpr int64_t SE3_ID = 3;
template <typename... Ts>
void lie_chk(int64_t group_id, const Tensor& first, const Ts&... rest) {
  TORCH_CHECK(group_id == SE3_ID, "lietorch_backends (devo_amd): only SE3 (group_id 3) is implemented, got ", group_id);
  for (const Tensor* t : {&first, &rest...}) {
    TORCH_CHECK(t->is_cuda(), "devo_amd: tensors must live on the GPU (the HIP path has no CPU fallback)");
    TORCH_CHECK(t->is_contiguous(), "lietorch_backends: input must be contiguous");                    // lietorch.cpp:7
    TORCH_CHECK(t->scalar_type() == at::kFloat || t->scalar_type() == at::kDouble, "lietorch_backends: float32/float64 only, got ", t->scalar_type());
    TORCH_CHECK(t->scalar_type() == first.scalar_type(), "lietorch_backends: mixed dtypes");
  }
}
typedef int (*un_fn)(const void*, void*, int64_t, int, devo_stream_t);
typedef int (*unb_fn)(const void*, const void*, void*, int64_t, int, devo_stream_t);
typedef int (*bin_fn)(const void*, const void*, void*, int64_t, int, devo_stream_t);
typedef int (*binb_fn)(const void*, const void*, const void*, void*, void*, int64_t, int, devo_stream_t);

template <un_fn F, int OUT>
Tensor lie_unary(int64_t gid, Tensor X) {
  lie_chk(gid, X);
  c10::DeviceGuard guard(X.device());
  Tensor out = at::empty({X.size(0), OUT}, X.options());
  check(F(X.data_ptr(), out.data_ptr(), X.size(0), dtype_code(X), stream_of(X)), "lietorch_backends");
  return out;
}
template <unb_fn F, int OUT>
TensorList lie_unary_bwd(int64_t gid, Tensor grad, Tensor X) {
  lie_chk(gid, grad, X);
  c10::DeviceGuard guard(X.device());
  Tensor out = at::empty({X.size(0), OUT}, X.options());
  check(F(grad.data_ptr(), X.data_ptr(), out.data_ptr(), X.size(0), dtype_code(X), stream_of(X)), "lietorch_backends");
  return {out};
}
template <bin_fn F, int OUT>
Tensor lie_binary(int64_t gid, Tensor X, Tensor y) {
  lie_chk(gid, X, y);
  c10::DeviceGuard guard(X.device());
  Tensor out = at::empty({X.size(0), OUT}, X.options());
  check(F(X.data_ptr(), y.data_ptr(), out.data_ptr(), X.size(0), dtype_code(X), stream_of(X)), "lietorch_backends");
  return out;
}
template <binb_fn F, int DY>
TensorList lie_binary_bwd(int64_t gid, Tensor grad, Tensor X, Tensor y) {
  lie_chk(gid, grad, X, y);
  c10::DeviceGuard guard(X.device());
  Tensor dX = at::empty({X.size(0), 7}, X.options()), dy = at::empty({X.size(0), DY}, X.options());
  check(F(grad.data_ptr(), X.data_ptr(), y.data_ptr(), dX.data_ptr(), dy.data_ptr(), X.size(0), dtype_code(X), stream_of(X)), "lietorch_backends");
  return {dX, dy};
}
Tensor lie_as_matrix(int64_t gid, Tensor X) {
  lie_chk(gid, X);
  c10::DeviceGuard guard(X.device());
  Tensor out = at::empty({X.size(0), 4, 4}, X.options());
  check(devo_se3_as_matrix(X.data_ptr(), out.data_ptr(), X.size(0), dtype_code(X), stream_of(X)), "devo_se3_as_matrix");
  return out;
}
Tensor lie_projector(int64_t, Tensor) {
  TORCH_CHECK(false, "lietorch_backends.projector (ToVec/FromVec) is outside the DEVO hot path (SURVEY.md §2.1 row 3): never reached from devo.py / enet.py / train.py");
  return Tensor();
}

// extras for devo_amd.backends.cuda_corr (one cache for both bindings)
std::tuple<Tensor, c10::optional<Tensor>, int64_t> corr_fast_layout(Tensor fmap2, int64_t n_edges, bool allow_split) {
  require_gpu(fmap2);
  c10::DeviceGuard guard(fmap2.device());
  const Level l = fast_level(fmap2, n_edges, allow_split);
  return {l.data, l.exps.defined() ? c10::optional<Tensor>(l.exps) : c10::nullopt, (int64_t)l.cblock};
}
c10::optional<Tensor> corr_patch_operand(Tensor fmap1) {
  require_gpu(fmap1);
  c10::DeviceGuard guard(fmap1.device());
  const Tensor t = patch_operand(fmap1.contiguous());
  return t.defined() ? c10::optional<Tensor>(t) : c10::nullopt;
}
void clear_caches() {
  g_levels.clear(); g_patches.clear(); g_cl.clear();
  g_last_plan = LastPlan();
  ba_prep_invalidate();
  { std::lock_guard<std::mutex> g(g_wr_mu); g_writes.clear(); }
  std::lock_guard<std::mutex> g(g_ws_mu);
  g_ws = Tensor(); g_ws_key = WsKey{-1, -1, -1, -1, nullptr};
}

// torch.ops forms of the in-place / list-returning functions
TensorList ba_forward_op(Tensor poses, Tensor patches, Tensor intrinsics, Tensor target, Tensor weight, Tensor lmbda, Tensor ii, Tensor jj, Tensor kk, int64_t t0, int64_t t1,
                         int64_t iterations) {
  return ba_forward(poses, patches, intrinsics, target, weight, lmbda, ii, jj, kk, t0, t1, iterations, c10::nullopt, c10::nullopt, false);
}

}  // namespace

// torch.ops.devo_hip.* (north_star's wording; the reference itself registers pybind modules only)
TORCH_LIBRARY(devo_hip, m) {
  m.def("corr_forward(Tensor fmap1, Tensor fmap2, Tensor coords, Tensor ii, Tensor jj, int radius) -> Tensor[]");
  m.def("corr_backward(Tensor fmap1, Tensor fmap2, Tensor coords, Tensor ii, Tensor jj, Tensor grad, int radius) -> Tensor[]");
  m.def("patchify_forward(Tensor net, Tensor coords, int radius) -> Tensor[]");
  m.def("patchify_backward(Tensor net, Tensor coords, Tensor gradient, int radius) -> Tensor[]");
  m.def("ba_forward(Tensor(a!) poses, Tensor(b!) patches, Tensor intrinsics, Tensor target, Tensor weight, Tensor lmbda, Tensor ii, Tensor jj, Tensor kk, int t0, int t1, int iterations) -> Tensor[]");
  m.def("ba_neighbors(Tensor ii, Tensor jj) -> Tensor[]");
  m.def("ba_reproject(Tensor poses, Tensor patches, Tensor intrinsics, Tensor ii, Tensor jj, Tensor kk) -> Tensor");
  m.def("se3_exp(int group_id, Tensor a) -> Tensor");
  m.def("se3_log(int group_id, Tensor X) -> Tensor");
  m.def("se3_inv(int group_id, Tensor X) -> Tensor");
  m.def("se3_mul(int group_id, Tensor X, Tensor Y) -> Tensor");
  m.def("se3_adj(int group_id, Tensor X, Tensor a) -> Tensor");
  m.def("se3_adjT(int group_id, Tensor X, Tensor a) -> Tensor");
  m.def("se3_act(int group_id, Tensor X, Tensor p) -> Tensor");
  m.def("se3_act4(int group_id, Tensor X, Tensor p) -> Tensor");
}
TORCH_LIBRARY_IMPL(devo_hip, CompositeExplicitAutograd, m) {
  m.impl("corr_forward", &corr_forward);
  m.impl("corr_backward", &corr_backward);
  m.impl("patchify_forward", &patchify_forward);
  m.impl("patchify_backward", &patchify_backward);
  m.impl("ba_forward", &ba_forward_op);
  m.impl("ba_neighbors", &ba_neighbors);
  m.impl("ba_reproject", &ba_reproject);
  m.impl("se3_exp", &lie_unary<devo_se3_exp, 7>);
  m.impl("se3_log", &lie_unary<devo_se3_log, 6>);
  m.impl("se3_inv", &lie_unary<devo_se3_inv, 7>);
  m.impl("se3_mul", &lie_binary<devo_se3_mul, 7>);
  m.impl("se3_adj", &lie_binary<devo_se3_adj, 6>);
  m.impl("se3_adjT", &lie_binary<devo_se3_adjT, 6>);
  m.impl("se3_act", &lie_binary<devo_se3_act, 3>);
  m.impl("se3_act4", &lie_binary<devo_se3_act4, 4>);
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "devo_amd._C: compiled binding of libdevo_hip.so with the reference's module interfaces";
  m.def("abi_version", []() { return devo_abi_version(); });
  m.def("clear_caches", &clear_caches, "drop the cached pyramid copies, patch operands and the BA workspace");
  namespace py = pybind11;

  auto corr = m.def_submodule("cuda_corr", "devo/altcorr/correlation.cpp:57-63");
  corr.def("forward", &corr_forward, "correlation.cpp:58");
  corr.def("backward", &corr_backward, "correlation.cpp:59");
  corr.def("patchify_forward", &patchify_forward, "correlation.cpp:61");
  corr.def("patchify_backward", &patchify_backward, "correlation.cpp:62");
  corr.def("last_backward_path", &corr_last_backward_path);
  corr.def("last_forward_path", [] {
    static const char* names[] = {"dense-product", "mfma4x4", "staged", "generic", "dense-product-groups"};
    const int p = devo_corr_forward_last_path();
    return std::string((p >= 0 && p < 5) ? names[p] : "");
  }, "the kernel the last forward lookup of this thread launched (devo_corr_forward_last_path)");
  corr.def("_fast_layout", &corr_fast_layout, py::arg("fmap2"), py::arg("n_edges"), py::arg("allow_split") = true);
  corr.def("_patch_operand", &corr_patch_operand);
  corr.def("_cached_levels", []() { std::lock_guard<std::mutex> g(g_levels.mu); return (int64_t)g_levels.items.size(); });
  corr.def("_is_tracked", [](int64_t ptr) { return is_tracked((const void*)(intptr_t)ptr); }, "does the cache hold a converted copy of the tensor at this address?");
  corr.def("_note_write", [](int64_t ptr, int64_t version_after, int64_t off, int64_t len) { note_write((const void*)(intptr_t)ptr, (uint32_t)version_after, off, len); },
           "a __setitem__ wrote elements [off, off + len) of the tensor at `ptr` and left its version counter at `version_after` (len < 0: region unknown)");
  corr.def("_convert_stats", [] {
    return std::make_tuple(g_conv.level_full, g_conv.level_frames, g_conv.patch_full, g_conv.patch_ranges, g_conv.patches_partial);
  }, "(whole levels converted, single frames converted, whole patch operands, patch ranges, patches in them) since the module was loaded");

  auto ba = m.def_submodule("cuda_ba", "devo/fastba/ba.cpp:152-157");
  ba.def("forward", &ba_forward, "ba.cpp:153 (in place, returns [])", py::arg("poses"), py::arg("patches"), py::arg("intrinsics"), py::arg("target"), py::arg("weight"),
         py::arg("lmbda"), py::arg("ii"), py::arg("jj"), py::arg("kk"), py::arg("t0"), py::arg("t1"), py::arg("iterations"), py::arg("ws") = py::none(),
         py::arg("status") = py::none(), py::arg("prepared") = false);
  ba.def("neighbors", &ba_neighbors, "ba.cpp:154");
  ba.def("_prep_invalidate", &ba_prep_invalidate, "forget the prepared index tables of the last forward() (an explicit prepare() rewrote the workspace)");
  ba.def("_prep_stats", [] { std::lock_guard<std::mutex> g(g_prep_mu); return std::make_pair(g_prep_hits, g_prep_misses); },
         "(hits, misses) of forward()'s prepared-table cache since the module was loaded");
  ba.def("last_path", [] {
    static const char* acc[] = {"register", "lds", "global"};
    static const char* sol[] = {"chain", "lds", "global"};
    const int p = devo_ba_last_path();
    return p < 0 ? std::string("") : std::string("accumulate:") + acc[p & 3] + " solve:" + sol[(p >> 2) & 3];
  }, "the kernels the last forward() of this thread ran (devo_ba_last_path)");
  ba.def("reproject", &ba_reproject, "ba.cpp:155");
  ba.def("transform_coords", &ba_transform, py::arg("poses"), py::arg("patches"), py::arg("intrinsics"), py::arg("ii"), py::arg("jj"), py::arg("kk"),
         py::arg("layout_2pp") = false);

  auto lie = m.def_submodule("lietorch_backends", "devo/lietorch/src/lietorch.cpp:286-316 (SE3)");
  lie.def("expm", &lie_unary<devo_se3_exp, 7>);
  lie.def("expm_backward", &lie_unary_bwd<devo_se3_exp_backward, 6>);
  lie.def("logm", &lie_unary<devo_se3_log, 6>);
  lie.def("logm_backward", &lie_unary_bwd<devo_se3_log_backward, 7>);
  lie.def("inv", &lie_unary<devo_se3_inv, 7>);
  lie.def("inv_backward", &lie_unary_bwd<devo_se3_inv_backward, 7>);
  lie.def("mul", &lie_binary<devo_se3_mul, 7>);
  lie.def("mul_backward", &lie_binary_bwd<devo_se3_mul_backward, 7>);
  lie.def("adj", &lie_binary<devo_se3_adj, 6>);
  lie.def("adj_backward", &lie_binary_bwd<devo_se3_adj_backward, 6>);
  lie.def("adjT", &lie_binary<devo_se3_adjT, 6>);
  lie.def("adjT_backward", &lie_binary_bwd<devo_se3_adjT_backward, 6>);
  lie.def("act", &lie_binary<devo_se3_act, 3>);
  lie.def("act_backward", &lie_binary_bwd<devo_se3_act_backward, 3>);
  lie.def("act4", &lie_binary<devo_se3_act4, 4>);
  lie.def("act4_backward", &lie_binary_bwd<devo_se3_act4_backward, 4>);
  lie.def("as_matrix", &lie_as_matrix);
  lie.def("projector", &lie_projector);
  lie.def("Jinv", &lie_binary<devo_se3_jinv, 6>);
}
