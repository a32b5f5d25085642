// Shared helpers for the gfx950 kernels of libmmtta.so.  CDNA4 only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/mmtta.h"

namespace mmtta {

constexpr int WAVE = 64;

void set_error(const char* fmt, ...);

#define MMTTA_CHECK(cond, code, ...)      \
  do {                                    \
    if (!(cond)) {                        \
      ::mmtta::set_error(__VA_ARGS__);    \
      return (code);                      \
    }                                     \
  } while (0)

inline int launch_status(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return MMTTA_ERR_LAUNCH;
  }
  return MMTTA_OK;
}

inline bool is_cl(const mmtta_tensor* t) { return t->sc == 1 || t->c == 1; }

inline bool is_f32(const mmtta_tensor* t) { return t->dtype == MMTTA_F32; }
inline bool is_bf16(const mmtta_tensor* t) { return t->dtype == MMTTA_BF16; }

// ---- storage-type helpers.  In `bf16` precision the forward activations (raw conv outputs, residual-unit outputs,
// concat buffers) are STORED as bf16 (what torch autocast does); gradients, logits, statistics and weights stay fp32.
// A tensor's base pointer is carried as `float*` either way; `bf` says the elements are 2 bytes wide, offsets are in
// ELEMENTS.  4 consecutive channels = one 8-byte (bf16) or 16-byte (fp32) access.
__device__ __forceinline__ float bf16_bits_to_f32(unsigned int h) { return __uint_as_float(h << 16); }
__device__ __forceinline__ unsigned int f32x2_to_bf16x2(float lo, float hi) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  bf2 v;
  v[0] = (__bf16)lo;   // v_cvt_pk_bf16_f32: round to nearest even, NaN preserved
  v[1] = (__bf16)hi;
  return __builtin_bit_cast(unsigned int, v);
}
__device__ __forceinline__ float4 ld4_any(const float* base, long long eoff, int bf) {
  if (bf) {
    const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + eoff);
    return make_float4(bf16_bits_to_f32(u.x & 0xffffu), bf16_bits_to_f32(u.x >> 16), bf16_bits_to_f32(u.y & 0xffffu),
                       bf16_bits_to_f32(u.y >> 16));
  }
  return *reinterpret_cast<const float4*>(base + eoff);
}
__device__ __forceinline__ void st4_any(float* base, long long eoff, float4 v, int bf) {
  if (bf) {
    uint2 u;
    u.x = f32x2_to_bf16x2(v.x, v.y);
    u.y = f32x2_to_bf16x2(v.z, v.w);
    *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(base) + eoff) = u;
  } else {
    *reinterpret_cast<float4*>(base + eoff) = v;
  }
}
__device__ __forceinline__ float ld1_any(const float* base, long long eoff, int bf) {
  if (bf) return bf16_bits_to_f32(reinterpret_cast<const unsigned short*>(base)[eoff]);
  return base[eoff];
}
__device__ __forceinline__ void st1_any(float* base, long long eoff, float v, int bf) {
  if (bf) reinterpret_cast<unsigned short*>(base)[eoff] = (unsigned short)(f32x2_to_bf16x2(v, 0.f) & 0xffffu);
  else base[eoff] = v;
}
// 8 consecutive channels: two 16-byte loads (fp32) or one (bf16)
__device__ __forceinline__ void ld8_any(const float* base, long long eoff, int bf, float4& lo, float4& hi) {
  if (bf) {
    const uint4 u = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(base) + eoff);
    lo = make_float4(bf16_bits_to_f32(u.x & 0xffffu), bf16_bits_to_f32(u.x >> 16), bf16_bits_to_f32(u.y & 0xffffu),
                     bf16_bits_to_f32(u.y >> 16));
    hi = make_float4(bf16_bits_to_f32(u.z & 0xffffu), bf16_bits_to_f32(u.z >> 16), bf16_bits_to_f32(u.w & 0xffffu),
                     bf16_bits_to_f32(u.w >> 16));
  } else {
    lo = *reinterpret_cast<const float4*>(base + eoff);
    hi = *reinterpret_cast<const float4*>(base + eoff + 4);
  }
}

// raw 8-channel items (converted when they are committed: a conversion right behind the load would wait for it)
template <bool BF> struct Oct8 { float4 lo, hi; };
template <> struct Oct8<true> { uint4 q; };
// `base` points at the batch item (wave-uniform), the offsets are ELEMENTS below 2^31 (checked on the host): the loads
// take the scalar-base + 32-bit-offset form, no 64-bit vector arithmetic
template <bool BF>
__device__ __forceinline__ Oct8<BF> oct8_ld(const float* base, unsigned eoff_lo, unsigned eoff_hi) {
  Oct8<BF> o;
  if constexpr (BF) {
    o.q = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(base) + eoff_lo);
  } else {
    o.lo = *reinterpret_cast<const float4*>(base + eoff_lo);
    o.hi = *reinterpret_cast<const float4*>(base + eoff_hi);
  }
  return o;
}
__device__ __forceinline__ void oct8_f8(const Oct8<false>& o, float (&v)[8]) {
  v[0] = o.lo.x; v[1] = o.lo.y; v[2] = o.lo.z; v[3] = o.lo.w; v[4] = o.hi.x; v[5] = o.hi.y; v[6] = o.hi.z; v[7] = o.hi.w;
}
__device__ __forceinline__ void oct8_f8(const Oct8<true>& o, float (&v)[8]) {
  v[0] = bf16_bits_to_f32(o.q.x & 0xffffu); v[1] = __uint_as_float(o.q.x & 0xffff0000u);
  v[2] = bf16_bits_to_f32(o.q.y & 0xffffu); v[3] = __uint_as_float(o.q.y & 0xffff0000u);
  v[4] = bf16_bits_to_f32(o.q.z & 0xffffu); v[5] = __uint_as_float(o.q.z & 0xffff0000u);
  v[6] = bf16_bits_to_f32(o.q.w & 0xffffu); v[7] = __uint_as_float(o.q.w & 0xffff0000u);
}


// 8 consecutive channels out: two 16-byte stores (fp32) or one (bf16); `base` points at the batch item, offsets in ELEMENTS
template <bool BF>
__device__ __forceinline__ void oct8_st(float* base, unsigned eoff, const float (&v)[8]) {
  if constexpr (BF) {
    uint4 q;
    q.x = f32x2_to_bf16x2(v[0], v[1]); q.y = f32x2_to_bf16x2(v[2], v[3]);
    q.z = f32x2_to_bf16x2(v[4], v[5]); q.w = f32x2_to_bf16x2(v[6], v[7]);
    *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(base) + eoff) = q;
  } else {
    *reinterpret_cast<float4*>(base + eoff) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(base + eoff + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
}
// base pointer of batch item n of a tensor whose elements are 2 (BF) or 4 bytes wide; `sn` in ELEMENTS
template <bool BF>
__device__ __forceinline__ const float* item_base(const float* p, long long n, long long sn) {
  return BF ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(p) + n * sn) : p + n * sn;
}

// compile-time loop: f(std::integral_constant<int, I>) for I in [I0, N) - the index is usable as a template argument
// and an array subscript that never becomes a run-time value (register arrays stay in registers)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

template <int I, int N, int STEP, class F>
__device__ __forceinline__ void static_for_step(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for_step<I + STEP, N, STEP>(f);
  }
}

// Compile-time forms.  A run-time `bf` test around a load inside an unrolled loop makes hipcc branch around every
// load and wait for it at the end of its branch (measured: the whole adaptation 25 % slower with run-time tests, for
// fp32 AND bf16 storage), so hot loops are instantiated per storage type and selected by ONE wave-uniform branch outside:
//   MMTTA_BF_DISPATCH(flag, NAME, { ... uses `constexpr bool NAME` ... })
template <bool BF> __device__ __forceinline__ float4 ld4_t(const float* base, long long eoff) { return ld4_any(base, eoff, BF ? 1 : 0); }
template <bool BF> __device__ __forceinline__ float ld1_t(const float* base, long long eoff) { return ld1_any(base, eoff, BF ? 1 : 0); }
template <bool BF> __device__ __forceinline__ void st4_t(float* base, long long eoff, float4 v) { st4_any(base, eoff, v, BF ? 1 : 0); }
template <bool BF> __device__ __forceinline__ void st1_t(float* base, long long eoff, float v) { st1_any(base, eoff, v, BF ? 1 : 0); }
template <bool BF> __device__ __forceinline__ void ld8_t(const float* base, long long eoff, float4& lo, float4& hi) {
  ld8_any(base, eoff, BF ? 1 : 0, lo, hi);
}
#define MMTTA_BF_DISPATCH(flag, NAME, ...) \
  do {                                      \
    if (flag) {                             \
      constexpr bool NAME = true;           \
      __VA_ARGS__                           \
    } else {                                \
      constexpr bool NAME = false;          \
      __VA_ARGS__                           \
    }                                       \
  } while (0)

// Device-side copy of a tensor view (fp32 data, or bf16 data when `bf`: see the storage helpers above).
struct TV {
  float* p;
  int n, c, d, h, w;
  long long sn, sc, sd, sh, sw;
  int flags;
  int bf;
};

inline TV tv(const mmtta_tensor* t) {
  TV v;
  v.p = (float*)t->ptr;
  v.flags = t->flags;
  v.bf = t->dtype == MMTTA_BF16 ? 1 : 0;
  v.n = t->n; v.c = t->c; v.d = t->d; v.h = t->h; v.w = t->w;
  v.sn = t->sn; v.sc = t->sc; v.sd = t->sd; v.sh = t->sh; v.sw = t->sw;
  return v;
}

// linear voxel index (n, z, y, x order) <-> coordinates / element offset of a view
__device__ __forceinline__ void vox_decompose(const TV& t, long long v, int& n, int& z, int& y, int& x) {
  x = (int)(v % t.w); v /= t.w;
  y = (int)(v % t.h); v /= t.h;
  z = (int)(v % t.d);
  n = (int)(v / t.d);
}
__device__ __forceinline__ long long vox_addr(const TV& t, int n, int z, int y, int x) {
  return (long long)n * t.sn + (long long)z * t.sd + (long long)y * t.sh + (long long)x * t.sw;
}

// Device-side norm-on-load descriptor.
struct NL {
  const float* mean;
  const float* rstd;
  const float* gamma;
  const float* beta;
  int relu;
  const float* scale;
  const float* shift;
};

inline NL nl(const mmtta_norm_on_load* t) {
  NL r;
  if (t == nullptr) { r.mean = r.rstd = r.gamma = r.beta = r.scale = r.shift = nullptr; r.relu = 0; return r; }
  r.mean = t->mean; r.rstd = t->rstd; r.gamma = t->gamma; r.beta = t->beta; r.relu = t->relu;
  r.scale = t->scale; r.shift = t->scale ? t->shift : nullptr;
  return r;
}

// Per-volume parameter sets (mmtta_param_sets): which set batch item n reads / writes.  Items [q * ips, (q + 1) * ips)
// share set q; set q lives (q / inner) outer strides + (q % inner) inner strides behind the base pointer.  All-zero
// strides = one set for the whole batch (the plain entry points).  n is workgroup-uniform everywhere it is used, so the
// two divisions are scalar work paid once per workgroup.
struct PSets {
  int ips, inner;
  long long packed_outer, packed_inner;   // bytes between packed weight images
  long long weight_outer, weight_inner;   // elements between the weight gradients (torch weight layout)
  long long bias_outer, bias_inner;       // elements between bias vectors / bias gradients
};

inline PSets psets(const mmtta_param_sets* g) {
  PSets r;
  if (g == nullptr) { r.ips = 1; r.inner = 1; r.packed_outer = r.packed_inner = r.weight_outer = r.weight_inner = r.bias_outer = r.bias_inner = 0; return r; }
  r.ips = g->items_per_set; r.inner = g->inner;
  r.packed_outer = g->packed_outer; r.packed_inner = g->packed_inner;
  r.weight_outer = g->weight_outer; r.weight_inner = g->weight_inner;
  r.bias_outer = g->bias_outer; r.bias_inner = g->bias_inner;
  return r;
}
inline int psets_validate(const mmtta_param_sets* g, int n) {
  if (g == nullptr) return MMTTA_OK;
  MMTTA_CHECK(g->items_per_set >= 1 && g->inner >= 1, MMTTA_ERR_INVALID, "param sets: items_per_set %d, inner %d (both >= 1)",
              g->items_per_set, g->inner);
  MMTTA_CHECK(n % g->items_per_set == 0, MMTTA_ERR_INVALID, "param sets: batch %d is no multiple of items_per_set %d", n,
              g->items_per_set);
  MMTTA_CHECK(g->packed_outer % 16 == 0 && g->packed_inner % 16 == 0 && g->weight_outer % 4 == 0 && g->weight_inner % 4 == 0 &&
                  g->bias_outer % 4 == 0 && g->bias_inner % 4 == 0,
              MMTTA_ERR_INVALID, "param sets: strides must keep 16-byte alignment");
  return MMTTA_OK;
}
__host__ __device__ __forceinline__ bool psets_on(const PSets& g) {
  return (g.packed_outer | g.packed_inner | g.weight_outer | g.weight_inner | g.bias_outer | g.bias_inner) != 0;
}
__device__ __forceinline__ int pset_of(const PSets& g, int n) { return g.ips == 1 ? n : n / g.ips; }
__device__ __forceinline__ long long pset_packed_bytes(const PSets& g, int q) {
  return g.inner == 1 ? q * g.packed_outer : (q / g.inner) * g.packed_outer + (q % g.inner) * g.packed_inner;
}
__device__ __forceinline__ long long pset_weight_elems(const PSets& g, int q) {
  return g.inner == 1 ? q * g.weight_outer : (q / g.inner) * g.weight_outer + (q % g.inner) * g.weight_inner;
}
__device__ __forceinline__ long long pset_bias_elems(const PSets& g, int q) {
  return g.inner == 1 ? q * g.bias_outer : (q / g.inner) * g.bias_outer + (q % g.inner) * g.bias_inner;
}
// base pointers of batch item n's set (null stays null)
template <class T>
__device__ __forceinline__ const T* pset_packed(const PSets& g, const T* base, int n) {
  return reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + pset_packed_bytes(g, pset_of(g, n)));
}
__device__ __forceinline__ const float* pset_bias(const PSets& g, const float* base, int n) {
  return base == nullptr ? nullptr : base + pset_bias_elems(g, pset_of(g, n));
}

// scale/shift of channel c of batch item n for a norm-on-load (identity when mean == nullptr)
__device__ __forceinline__ void nl_coeff(const NL& t, int n, int C, int c, float& sc, float& sh) {
  if (t.scale != nullptr) { sc = t.scale[n * C + c]; sh = t.shift[n * C + c]; return; }   // precombined: two loads
  if (t.mean == nullptr) { sc = 1.f; sh = 0.f; return; }
  float mu = t.mean[n * C + c], rs = t.rstd[n * C + c];
  float g = t.gamma ? t.gamma[c] : 1.f;
  float b = t.beta ? t.beta[c] : 0.f;
  sc = rs * g;
  sh = b - mu * sc;
}

// J consecutive channels c0 .. c0+J-1 (channels >= C get 0, 0).  All loads are issued straight-line from clamped
// indices under wave-uniform branches only: a load inside a per-channel branch gets its own s_waitcnt and the J
// round trips serialise (measured: 8 channels = ~16 exposed L2 latencies per staging pass).
template <int J>
__device__ __forceinline__ void nl_coeff_vec(const NL& t, int n, int C, int c0, float* sc, float* sh) {
  if (t.scale != nullptr) {
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int i = n * C + min(c0 + j, C - 1);
      sc[j] = t.scale[i]; sh[j] = t.shift[i];
    }
  } else if (t.mean == nullptr) {
#pragma unroll
    for (int j = 0; j < J; ++j) { sc[j] = 1.f; sh[j] = 0.f; }
  } else {
    float mu[J], rs[J], g[J], b[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int cc = min(c0 + j, C - 1);
      mu[j] = t.mean[n * C + cc]; rs[j] = t.rstd[n * C + cc];
      g[j] = 1.f; b[j] = 0.f;
    }
    if (t.gamma != nullptr) {
#pragma unroll
      for (int j = 0; j < J; ++j) g[j] = t.gamma[min(c0 + j, C - 1)];
    }
    if (t.beta != nullptr) {
#pragma unroll
      for (int j = 0; j < J; ++j) b[j] = t.beta[min(c0 + j, C - 1)];
    }
#pragma unroll
    for (int j = 0; j < J; ++j) { sc[j] = rs[j] * g[j]; sh[j] = b[j] - mu[j] * sc[j]; }
  }
#pragma unroll
  for (int j = 0; j < J; ++j)
    if (c0 + j >= C) { sc[j] = 0.f; sh[j] = 0.f; }
}

__device__ __forceinline__ float nl_apply(float x, float sc, float sh, int relu) {
  float v = fmaf(x, sc, sh);
  return relu ? fmaxf(v, 0.f) : v;
}

// Workgroups are dealt round-robin over the 8 XCDs, each with its own L2 (observed placement: a speed matter only).
// Give the workgroups that share an XCD one CONTIGUOUS range of logical ids, so that neighbours in the logical order
// (which share halo rows / operand panels) hit the same L2.  Bijective for any workgroup count.
__device__ __forceinline__ unsigned xcd_contiguous_id(unsigned b, unsigned nwg) {
  const unsigned x = b & 7u, i = b >> 3, q = nwg >> 3, r = nwg & 7u;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// [first, last) of the work units of logical workgroup lb when `units` are split into nwg contiguous ranges
__device__ __forceinline__ void unit_range(long long units, unsigned lb, unsigned nwg, long long& first, long long& last) {
  const long long per = (units + nwg - 1) / nwg;
  first = per * lb;
  last = first + per < units ? first + per : units;
}

// Sum over the 64 lanes, returned to every lane.  Six DPP adds (row_shr 1 / 2 / 4 / 8 inside the rows of 16, then
// row_bcast 15 and 31 across rows: the total lands in lane 63) and one readlane, instead of six ds_bpermute round trips
// through the LDS crossbar: the tiny weight gradient reduces 84 values per wave (a quarter of its vector instructions).
__device__ __forceinline__ float wave_sum(float v) {
  auto step = [](float x, auto ctrl, auto rmask) {
    const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, decltype(rmask)::value, 0xf, true);
    return x + __builtin_bit_cast(float, moved);
  };
  v = step(v, std::integral_constant<int, 0x111>{}, std::integral_constant<int, 0xf>{});   // row_shr:1
  v = step(v, std::integral_constant<int, 0x112>{}, std::integral_constant<int, 0xf>{});   // row_shr:2
  v = step(v, std::integral_constant<int, 0x114>{}, std::integral_constant<int, 0xf>{});   // row_shr:4
  v = step(v, std::integral_constant<int, 0x118>{}, std::integral_constant<int, 0xf>{});   // row_shr:8 -> lane 15 of a row = row total
  v = step(v, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{});   // row_bcast:15 into rows 1, 3
  v = step(v, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{});   // row_bcast:31 into rows 2, 3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ------------------------------------------------------------------ optimizer arithmetic (one element)
struct OptimArgs {
  float lr, beta1, beta2, eps, wd, momentum, dampening;
  int nesterov;
};

// Every product-sum below is an EXPLICIT fused multiply-add and implicit contraction is switched off: the function is
// inlined into several kernels (the arena optimizer, the fused weight-gradient reductions) and hipcc's contraction choices
// differ with the surrounding code - left to itself the two gave moments one ulp apart from the second step on.
template <int KIND>
__device__ __forceinline__ void optim_update(float& pi, float gi, float& mi, float& vi, bool decay, const OptimArgs& a,
                                             float step_size, float bc2_sqrt, bool first) {
#pragma clang fp contract(off)
  if (KIND == 0) {
    if (decay) gi = fmaf(a.wd, pi, gi);
    mi = fmaf(1.f - a.beta1, gi - mi, mi);
    vi = fmaf((1.f - a.beta2) * gi, gi, vi * a.beta2);
    const float denom = sqrtf(vi) / bc2_sqrt + a.eps;
    pi = fmaf(-step_size, mi / denom, pi);
  } else if (KIND == 1) {
    if (decay) pi = pi * (1.f - a.lr * a.wd);
    mi = fmaf(1.f - a.beta1, gi - mi, mi);
    vi = fmaf((1.f - a.beta2) * gi, gi, vi * a.beta2);
    const float denom = sqrtf(vi) / bc2_sqrt + a.eps;
    pi = fmaf(-step_size, mi / denom, pi);
  } else {
    if (decay) gi = fmaf(a.wd, pi, gi);
    if (a.momentum != 0.f) {
      mi = first ? gi : fmaf(a.momentum, mi, (1.f - a.dampening) * gi);
      gi = a.nesterov ? fmaf(a.momentum, mi, gi) : mi;
    }
    pi = fmaf(-a.lr, gi, pi);
  }
}

// bias-corrected step size and sqrt(1 - beta2^t) of optimizer step t0 + 1 (double arithmetic, as torch does on the host)
__device__ __forceinline__ void optim_scalars(int kind, const OptimArgs& a, int t0, float& step_size, float& bc2_sqrt) {
  const double t = (double)(t0 + 1);
  const double bc1 = 1.0 - pow((double)a.beta1, t);
  const double bc2 = 1.0 - pow((double)a.beta2, t);
  step_size = kind == 2 ? a.lr : (float)((double)a.lr / bc1);
  bc2_sqrt = kind == 2 ? 1.f : (float)sqrt(bc2);
}

// shared between translation units (defined in pointwise.hip)
int channel_partial_rows(const mmtta_tensor* t);                        // partial rows per batch item
int launch_channel_sums(const mmtta_tensor* x, float* part, hipStream_t s);  // part: [N*rows][2][C] (sum, sumsq)

// direct path for layers producing <= 4 channels (conv_direct.hip)
extern int g_profile_main_only;     // api.hip: mmtta_set_option(MMTTA_OPT_PROFILE_MAIN_KERNEL_ONLY)
extern int g_igemm_pipeline;        // api.hip: MMTTA_OPT_IGEMM_PIPELINE
extern int g_wgrad_vec;             // api.hip: MMTTA_OPT_WGRAD_VECTOR_STAGING
extern int g_igemm_lean;            // api.hip: MMTTA_OPT_IGEMM_LEAN
extern int g_cls_fused_min;         // api.hip: MMTTA_OPT_CLASS_FUSED_MIN_WORKGROUPS
extern int g_thin_mfma;             // api.hip: MMTTA_OPT_THIN_MFMA
extern int g_epilogue_vec;          // api.hip: MMTTA_OPT_EPILOGUE_VEC16
extern int g_tune[4];               // api.hip: launch-geometry knobs (MMTTA_OPT_SPLITK_BELOW ... MMTTA_OPT_WGRAD_THIN_SLABS)
bool direct_applicable(const mmtta_conv_desc* d);
// bytes of the fragment-ordered bf16 weight image W' of the 2x2x2 gather-GEMM up-convolution (conv_direct.hip,
// upconv8_kernel), stored behind the fp32 tap image of the packed buffer; 0 for every other layer
long long upconv8_image_bytes(const mmtta_conv_desc* d);
// bytes of the thin-K convolution's B-fragment image behind the fp32 tap image (conv_direct.hip: chan_mfma_kernel), 0 if none
long long chan_frag_bytes(const mmtta_conv_desc* d);
int direct_blocks_per_n(const mmtta_conv_desc* d, const mmtta_tensor* x, const mmtta_tensor* y);
bool pointwise_small_applicable(const mmtta_conv_desc* d, const mmtta_tensor* x, const mmtta_tensor* y, const float* stats,
                                const mmtta_conv_epilogue* epi, const mmtta_norm_on_load* x_norm);
int pointwise_small_run(const mmtta_tensor* x, const void* packed, int Kp, int Np, const float* bias, const mmtta_tensor* y,
                        int accumulate, const PSets& sets, hipStream_t stream);
bool chan_applicable(const mmtta_conv_desc* d, const mmtta_tensor* x, const mmtta_tensor* y);
int chan_tiles_per_n(const mmtta_tensor* y);
int chan_conv_run(const mmtta_conv_desc* d, const mmtta_tensor* x, const mmtta_norm_on_load* x_norm, const void* packed, int Kp,
                  int Np, const float* bias, const mmtta_conv_epilogue* epi, const mmtta_tensor* y, int accumulate, float* stats,
                  const PSets& sets, hipStream_t stream);
int direct_conv_run(const mmtta_conv_desc* d, const mmtta_tensor* x, const mmtta_norm_on_load* x_norm, const void* packed,
                    const float* bias, const mmtta_conv_epilogue* epi, const mmtta_tensor* y, int accumulate, float* stats,
                    const PSets& sets, hipStream_t stream);

}  // namespace mmtta
