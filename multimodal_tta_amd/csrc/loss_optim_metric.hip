// Entropy objective (+gradient), fused Adam over the flat parameter arena, and the
// sigmoid -> threshold -> Dice counting tail (gfx950).  All three are single-pass HBM-bound
// kernels; see include/mmtta.h for the reference lines each one stands in for.
#include "common.h"

namespace mmtta {

// ------------------------------------------------------------------ entropy loss
constexpr int ENT_MAX_BLOCKS = 2048;
constexpr int ENT_MAX_R = 16;

__device__ __forceinline__ double block_sum_d(double v, double* sh) {
  v = wave_sum_d(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0)
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sh[w];
  return t;  // valid on thread 0
}

__global__ __launch_bounds__(256) void entropy_bernoulli_kernel(TV z, TV dz, double* partial, float inv_count, int per_item) {
  __shared__ double sh[4];
  if (per_item) {      // N independent volumes: this workgroup column works on batch item blockIdx.y alone
    z.p += (long long)blockIdx.y * z.sn; dz.p += (long long)blockIdx.y * dz.sn;
    z.n = 1; dz.n = 1;
    partial += (long long)blockIdx.y * gridDim.x;
  }
  const int C = z.c;
  const long long total = (long long)z.n * z.d * z.h * z.w * C;
  double acc = 0.0;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long long v = i / C;
    const int x = (int)(v % z.w); v /= z.w;
    const int y = (int)(v % z.h); v /= z.h;
    const int zz = (int)(v % z.d);
    const int n = (int)(v / z.d);
    const float t = z.p[n * z.sn + zz * z.sd + y * z.sh + x * z.sw + c];
    const float e = expf(-fabsf(t));
    const float sig = t >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
    const float softplus = fmaxf(t, 0.f) + log1pf(e);
    acc += (double)(softplus - t * sig);
    dz.p[n * dz.sn + zz * dz.sd + y * dz.sh + x * dz.sw + c] = -t * sig * (1.f - sig) * inv_count;
  }
  const double t = block_sum_d(acc, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

// Fast path of the Bernoulli objective: <= 4 regions in 16-byte voxel rows, dense voxel order.  A thread owns a
// voxel: one 16-byte load, one 16-byte store (the gradient tensor owns its pad lane), no index arithmetic.
// The kernel ran at 100 - 120 % vector-ALU busy (profiles/r03d_sq_counters.md: 158 vector instructions per logit in
// libm's expf / log1pf / IEEE division and a double add per logit) - 203 us per group of 8 volumes at 128^3 against 85 us
// of HBM time, and vector issue is what the other lanes' kernels run short of.  Now per logit: v_exp_f32, v_rcp_f32,
// v_log_f32 and ~15 plain instructions; log1p(e) of e = exp(-|t|) in (0, 1] is the 4-term series below 2^-6 (relative
// error < 2e-8) and log(1 + e) above (absolute rounding 6e-8 against a value >= 0.0155); the voxel's <= 4 terms are summed
// in fp32 and enter the double accumulator once.  Against the libm form: loss within 1e-6 relative, gradient within 2e-6
// of its maximum (tests/test_hip_pointwise.py::test_entropy_loss holds both to the torch reference at 2e-5).
__device__ __forceinline__ void bernoulli_entropy_terms(float t, float& h, float& g) {
  const float a = fabsf(t);
  const float e = __builtin_amdgcn_exp2f(-a * 1.4426950408889634f);          // exp(-|t|)
  const float r = __builtin_amdgcn_rcpf(1.f + e);
  const float sig = t >= 0.f ? r : e * r;
  const float series = e * fmaf(e, fmaf(e, fmaf(e, -0.25f, 0.33333334f), -0.5f), 1.f);
  const float lg = __builtin_amdgcn_logf(1.f + e) * 0.6931471805599453f;     // v_log_f32 is log2
  const float l1p = e < 0.015625f ? series : lg;
  h = fmaxf(t, 0.f) + l1p - t * sig;
  g = -t * (e * r * r);            // sig (1 - sig) = sig(|t|) (1 - sig(|t|)) = r * (e r): even in t
}
// OBF: the gradient is bf16-stored (8-byte voxels; method.grad_storage - its readers round it to bf16 while staging)
template <bool OBF>
__global__ __launch_bounds__(256) void entropy_bernoulli_vec_kernel(TV z, TV dz, double* partial, float inv_count, int per_item) {
  __shared__ double sh[4];
  if (per_item) {      // N independent volumes: this workgroup column works on batch item blockIdx.y alone
    z.p += (long long)blockIdx.y * z.sn;
    dz.p = OBF ? reinterpret_cast<float*>(reinterpret_cast<unsigned short*>(dz.p) + (long long)blockIdx.y * dz.sn)
               : dz.p + (long long)blockIdx.y * dz.sn;
    z.n = 1; dz.n = 1;
    partial += (long long)blockIdx.y * gridDim.x;
  }
  const int C = z.c;
  const long long dhw = (long long)z.d * z.h * z.w;
  const long long total = (long long)z.n * dhw;
  double acc = 0.0;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    // one volume per launch in the adaptation loop: no 64-bit division per voxel then (uniform branch)
    const long long n = z.n == 1 ? 0 : i / dhw, v = i - n * dhw;
    const float4 t4 = *reinterpret_cast<const float4*>(z.p + n * z.sn + v * 4);
    const float ts[4] = {t4.x, t4.y, t4.z, t4.w};
    float g[4] = {0.f, 0.f, 0.f, 0.f};
    float h = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (c < C) {
        float hc, gc;
        bernoulli_entropy_terms(ts[c], hc, gc);
        h += hc;
        g[c] = gc * inv_count;
      }
    }
    acc += (double)h;
    st4_any(dz.p, n * dz.sn + v * 4, make_float4(g[0], g[1], g[2], g[3]), OBF);
  }
  const double t = block_sum_d(acc, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

__global__ __launch_bounds__(256) void entropy_categorical_kernel(TV z, TV dz, double* partial, float inv_count, int per_item) {
  __shared__ double sh[4];
  if (per_item) {      // N independent volumes: this workgroup column works on batch item blockIdx.y alone
    z.p += (long long)blockIdx.y * z.sn; dz.p += (long long)blockIdx.y * dz.sn;
    z.n = 1; dz.n = 1;
    partial += (long long)blockIdx.y * gridDim.x;
  }
  const int R = z.c;
  const long long total = (long long)z.n * z.d * z.h * z.w;
  double acc = 0.0;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    long long v = i;
    const int x = (int)(v % z.w); v /= z.w;
    const int y = (int)(v % z.h); v /= z.h;
    const int zz = (int)(v % z.d);
    const int n = (int)(v / z.d);
    const float* zp = z.p + n * z.sn + zz * z.sd + y * z.sh + x * z.sw;
    float* gp = dz.p + n * dz.sn + zz * dz.sd + y * dz.sh + x * dz.sw;
    float t[ENT_MAX_R];
    float m = -INFINITY;
#pragma unroll
    for (int r = 0; r < ENT_MAX_R; ++r)
      if (r < R) { t[r] = zp[r]; m = fmaxf(m, t[r]); }
    float se = 0.f;
#pragma unroll
    for (int r = 0; r < ENT_MAX_R; ++r)
      if (r < R) se += expf(t[r] - m);
    const float lse = m + logf(se);
    float pz = 0.f;
#pragma unroll
    for (int r = 0; r < ENT_MAX_R; ++r)
      if (r < R) pz += expf(t[r] - lse) * t[r];
    const float H = lse - pz;
    acc += (double)H;
#pragma unroll
    for (int r = 0; r < ENT_MAX_R; ++r)
      if (r < R) {
        const float logp = t[r] - lse;
        gp[r] = -expf(logp) * (logp + H) * inv_count;
      }
  }
  const double t = block_sum_d(acc, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

__global__ __launch_bounds__(64) void entropy_finish_kernel(const double* partial, int nblocks, double inv_count,
                                                            float* loss) {
  partial += (long long)blockIdx.x * nblocks;      // one workgroup per independent item (a single one otherwise)
  double s = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 64) s += partial[i];
  s = wave_sum_d(s);
  if (threadIdx.x == 0) loss[blockIdx.x] = (float)(s * inv_count);
}

static int entropy_blocks(const mmtta_tensor* z) {
  const long long total = (long long)z->n * z->d * z->h * z->w * z->c;
  long long b = (total + 255) / 256;
  if (b < 1) b = 1;
  if (b > ENT_MAX_BLOCKS) b = ENT_MAX_BLOCKS;
  return (int)b;
}

// ------------------------------------------------------------------ fused optimizers over the flat arena
// KIND 0: torch.optim.Adam (coupled L2), 1: torch.optim.AdamW (decoupled decay), 2: torch.optim.SGD (momentum, dampening,
// nesterov).  One thread owns 4 consecutive parameters (16-byte accesses; the arena keeps every parameter and both
// segment boundaries 16-byte aligned), the decay / no-decay split is the segment boundary n_decay
// (reference src/core/experiment_manager.py:199-237).
// OptimArgs / optim_update / optim_scalars: common.h

template <int KIND>
__global__ __launch_bounds__(256) void optim_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, long long n,
                                                    long long n_decay, OptimArgs a, const int* __restrict__ step,
                                                    long long set_stride) {
  __shared__ float s_step_size, s_bc2_sqrt;
  {  // replica blockIdx.y of the arena (a group of volumes, each with its own parameters and optimizer state)
    const long long o = (long long)blockIdx.y * set_stride;
    p += o; g += o; m += o;
    if (v != nullptr) v += o;
  }
  const int t0 = *step;
  if (threadIdx.x == 0) optim_scalars(KIND, a, t0, s_step_size, s_bc2_sqrt);
  __syncthreads();
  const float step_size = s_step_size, bc2_sqrt = s_bc2_sqrt;
  const bool first = t0 == 0;
  const bool wd_on = a.wd != 0.f;
  const long long n4 = n >> 2;
  for (long long q = blockIdx.x * (long long)blockDim.x + threadIdx.x; q < n4; q += (long long)gridDim.x * blockDim.x) {
    const long long i = q << 2;
    float4 pq = *reinterpret_cast<const float4*>(p + i);
    const float4 gq = *reinterpret_cast<const float4*>(g + i);
    // step 0 starts from zero moments whatever the buffers hold (torch creates them as zeros): the episodic reset need not
    // clear them, and the first step does not read them (8 of its 28 bytes per parameter)
    float4 mq = make_float4(0.f, 0.f, 0.f, 0.f), vq = mq;
    if (!first) {
      if (KIND != 2 || a.momentum != 0.f) mq = *reinterpret_cast<const float4*>(m + i);
      if (KIND != 2) vq = *reinterpret_cast<const float4*>(v + i);
    }
    const bool decay = wd_on && i < n_decay;          // n_decay is a multiple of 4
    optim_update<KIND>(pq.x, gq.x, mq.x, vq.x, decay, a, step_size, bc2_sqrt, first);
    optim_update<KIND>(pq.y, gq.y, mq.y, vq.y, decay, a, step_size, bc2_sqrt, first);
    optim_update<KIND>(pq.z, gq.z, mq.z, vq.z, decay, a, step_size, bc2_sqrt, first);
    optim_update<KIND>(pq.w, gq.w, mq.w, vq.w, decay, a, step_size, bc2_sqrt, first);
    *reinterpret_cast<float4*>(p + i) = pq;
    if (KIND != 2 || a.momentum != 0.f) *reinterpret_cast<float4*>(m + i) = mq;
    if (KIND != 2) *reinterpret_cast<float4*>(v + i) = vq;
  }
  // ragged tail (n not a multiple of 4: never the case for the arena, kept for arbitrary callers)
  const long long i = (n4 << 2) + blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i < n) {
    float pi = p[i], mi = first ? 0.f : m[i], vi = (KIND != 2 && !first) ? v[i] : 0.f;
    optim_update<KIND>(pi, g[i], mi, vi, wd_on && i < n_decay, a, step_size, bc2_sqrt, first);
    p[i] = pi; m[i] = mi;
    if (KIND != 2) v[i] = vi;
  }
}

__global__ void step_inc_kernel(int* step) { *step += 1; }

// ------------------------------------------------------------------ mask + Dice counts
__global__ __launch_bounds__(256) void dice_counts_kernel(TV z, TV lab, float thr, unsigned long long* counts,
                                                          unsigned char* mask, int blocks_per_n) {
  __shared__ unsigned int sh[3][256];
  const int R = z.c;
  int rp = 1;
  while (rp < R) rp <<= 1;        // R <= 256 enforced by the host
  const int nvl = 256 / rp;
  const int r = threadIdx.x % rp, vl = threadIdx.x / rp;
  const int n = blockIdx.x / blocks_per_n, bn = blockIdx.x % blocks_per_n;
  const long long dhw = (long long)z.d * z.h * z.w;
  unsigned int ci = 0, cp = 0, cg = 0;
  if (r < R) {
    for (long long v = (long long)bn * nvl + vl; v < dhw; v += (long long)blocks_per_n * nvl) {
      long long t = v;
      const int x = (int)(t % z.w); t /= z.w;
      const int y = (int)(t % z.h);
      const int zz = (int)(t / z.h);
      const float lg = z.p[n * z.sn + (long long)r * z.sc + zz * z.sd + y * z.sh + x * z.sw];
      const float prob = 1.f / (1.f + expf(-lg));
      const unsigned int pb = prob >= thr ? 1u : 0u;
      const float lv = lab.p[n * lab.sn + (long long)r * lab.sc + zz * lab.sd + y * lab.sh + x * lab.sw];
      const unsigned int gb = lv > 0.5f ? 1u : 0u;
      ci += pb & gb; cp += pb; cg += gb;
      if (mask) mask[((long long)n * R + r) * dhw + v] = (unsigned char)pb;
    }
  }
  sh[0][threadIdx.x] = ci; sh[1][threadIdx.x] = cp; sh[2][threadIdx.x] = cg;
  __syncthreads();
  if (vl == 0 && r < R) {
    unsigned long long ti = 0, tp = 0, tg = 0;
    for (int j = 0; j < nvl; ++j) { ti += sh[0][j * rp + r]; tp += sh[1][j * rp + r]; tg += sh[2][j * rp + r]; }
    unsigned long long* c = counts + ((long long)n * R + r) * 3;
    atomicAdd(c + 0, ti); atomicAdd(c + 1, tp); atomicAdd(c + 2, tg);
  }
}

// ------------------------------------------------------------------ DiceCE sums (evaluation report_loss)
// per (n,r): sum p*y, sum p (or p^2), sum y (or y^2), p = sigmoid(z); plus the CE numerator:
// R == 1: BCE-with-logits (pos_weight);  R > 1: soft-label softmax CE  -sum_r w_r y_r log_softmax(z)_r.
// Block partials [N][blocks_per_n][R*3 + 1] in the scratch buffer, summed in block order by dice_ce_finish_kernel:
// no atomics, the sums (and the gradient built from them) are bitwise reproducible.
constexpr int DCE_MAX_R = 8;
__global__ __launch_bounds__(256) void dice_ce_sums_kernel(TV z, TV lab, const float* weight, int squared, int softmax,
                                                           double* out, int blocks_per_n) {
  __shared__ double sh[4];
  const int R = z.c;
  const int n = blockIdx.x / blocks_per_n, bn = blockIdx.x % blocks_per_n;
  const long long dhw = (long long)z.d * z.h * z.w;
  double acc[DCE_MAX_R * 3 + 1];
#pragma unroll
  for (int i = 0; i < DCE_MAX_R * 3 + 1; ++i) acc[i] = 0.0;
  for (long long v = (long long)bn * 256 + threadIdx.x; v < dhw; v += (long long)blocks_per_n * 256) {
    long long t = v;
    const int x = (int)(t % z.w); t /= z.w;
    const int y = (int)(t % z.h);
    const int zz = (int)(t / z.h);
    float lg[DCE_MAX_R], yv[DCE_MAX_R];
    float m = -INFINITY;
#pragma unroll
    for (int r = 0; r < DCE_MAX_R; ++r)
      if (r < R) {
        lg[r] = z.p[n * z.sn + (long long)r * z.sc + zz * z.sd + y * z.sh + x * z.sw];
        yv[r] = lab.p[n * lab.sn + (long long)r * lab.sc + zz * lab.sd + y * lab.sh + x * lab.sw];
        m = fmaxf(m, lg[r]);
      }
    float se = 0.f;
#pragma unroll
    for (int r = 0; r < DCE_MAX_R; ++r)
      if (r < R) se += expf(lg[r] - m);
    const float lse = m + logf(se);
#pragma unroll
    for (int r = 0; r < DCE_MAX_R; ++r)
      if (r < R) {
        const float e = expf(-fabsf(lg[r]));
        // Dice probability: sigmoid (multilabel heads) or softmax over the channels (softmax heads with more than one channel)
        const float p = (softmax && R > 1) ? expf(lg[r] - lse) : (lg[r] >= 0.f ? 1.f / (1.f + e) : e / (1.f + e));
        acc[r * 3 + 0] += (double)(p * yv[r]);
        acc[r * 3 + 1] += (double)(squared ? p * p : p);
        acc[r * 3 + 2] += (double)(squared ? yv[r] * yv[r] : yv[r]);
        if (R == 1) {
          const float pw = weight ? weight[0] : 1.f;
          const float softplus_neg = log1pf(e) + fmaxf(-lg[r], 0.f);
          acc[DCE_MAX_R * 3] += (double)((1.f - yv[r]) * lg[r] + (1.f + (pw - 1.f) * yv[r]) * softplus_neg);
        } else {
          const float wr = weight ? weight[r] : 1.f;
          acc[DCE_MAX_R * 3] += (double)(-wr * yv[r] * (lg[r] - lse));
        }
      }
  }
  for (int i = 0; i < DCE_MAX_R * 3 + 1; ++i) {
    const bool used = (i == DCE_MAX_R * 3) || (i / 3 < R);
    if (!used) continue;                       // uniform across the block
    const double t = block_sum_d(acc[i], sh);
    if (threadIdx.x == 0) {
      const int col = (i == DCE_MAX_R * 3) ? R * 3 : i;
      out[((long long)n * blocks_per_n + bn) * (R * 3 + 1) + col] = t;     // block partial: no atomics, fixed order later
    }
    __syncthreads();
  }
}

// out[n][col] = sum over the block partials of batch item n, in block order (deterministic)
__global__ __launch_bounds__(64) void dice_ce_finish_kernel(const double* part, int blocks_per_n, int cols, double* out) {
  const int n = blockIdx.x / cols, col = blockIdx.x % cols;
  double s = 0.0;
  for (int b = threadIdx.x; b < blocks_per_n; b += 64) s += part[((long long)n * blocks_per_n + b) * cols + col];
  s = wave_sum_d(s);
  if (threadIdx.x == 0) out[(long long)n * cols + col] = s;
}

// Gradient of DiceCE (lambda_dice * Dice + lambda_ce * CE, reduction mean) with respect to the logits, from the
// per-(n,r) sums of dice_ce_sums_kernel (monai DiceCELoss as built at reference
// src/core/trainers/seg_trainer.py:59-79).  One thread per voxel, all R channels.
struct DceGradArgs {
  TV z, lab, dz;
  const float* weight;
  const double* sums;       // [N][R*3+1]
  int squared, jaccard, include_background, softmax;
  float lambda_dice, lambda_ce, smooth_nr, smooth_dr;
};

__global__ __launch_bounds__(256) void dice_ce_grad_kernel(DceGradArgs a) {
  const int R = a.z.c;
  const long long dhw = (long long)a.z.d * a.z.h * a.z.w;
  const long long total = (long long)a.z.n * dhw;
  const int r0 = (!a.include_background && R > 1) ? 1 : 0;
  const int Rd = R - r0;                                  // channels in the Dice mean
  const float dice_scale = a.lambda_dice / (float)((long long)a.z.n * Rd);
  const float ce_scale = a.lambda_ce / (float)((long long)a.z.n * dhw * (R == 1 ? 1 : 1));
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(i / dhw);
    long long t = i - (long long)n * dhw;
    const int x = (int)(t % a.z.w); t /= a.z.w;
    const int y = (int)(t % a.z.h);
    const int zz = (int)(t / a.z.h);
    float lg[DCE_MAX_R], yv[DCE_MAX_R], g[DCE_MAX_R];
    float m = -INFINITY;
#pragma unroll
    for (int r = 0; r < DCE_MAX_R; ++r)
      if (r < R) {
        lg[r] = a.z.p[n * a.z.sn + (long long)r * a.z.sc + zz * a.z.sd + y * a.z.sh + x * a.z.sw];
        yv[r] = a.lab.p[n * a.lab.sn + (long long)r * a.lab.sc + zz * a.lab.sd + y * a.lab.sh + x * a.lab.sw];
        m = fmaxf(m, lg[r]);
      }
    float se = 0.f, S = 0.f;
#pragma unroll
    for (int r = 0; r < DCE_MAX_R; ++r)
      if (r < R) { se += expf(lg[r] - m); S += (a.weight ? a.weight[r] : 1.f) * yv[r]; }
    const bool smx = a.softmax && R > 1;
    // softmax heads: dDice/dz_k = p_k (dLdp_k - sum_r dLdp_r p_r) (the softmax Jacobian); first pass: dLdp_r and their p-weighted sum
    float dldp[DCE_MAX_R], dot = 0.f;
#pragma unroll
    for (int r = 0; r < DCE_MAX_R; ++r) dldp[r] = 0.f;
#pragma unroll
    for (int r = 0; r < DCE_MAX_R; ++r)
      if (r < R) {
        const float e = expf(-fabsf(lg[r]));
        const float p = smx ? expf(lg[r] - m) / se : (lg[r] >= 0.f ? 1.f / (1.f + e) : e / (1.f + e));
        float gd = 0.f;
        if (r >= r0) {
          const double* sm = a.sums + (long long)n * (R * 3 + 1) + r * 3;
          const float I = (float)sm[0], Q = (float)sm[1], G = (float)sm[2];
          const float A = 2.f * I + a.smooth_nr;
          float den = G + Q;
          float dden = a.squared ? 2.f * p : 1.f;            // d(sum p or p^2)/dp
          if (a.jaccard) { den = 2.f * (den - I); dden = 2.f * (dden - yv[r]); }
          const float Dn = den + a.smooth_dr;
          const float dfdp = -(2.f * yv[r] * Dn - A * dden) / (Dn * Dn);
          // class weights scale the Dice terms only when there is more than one Dice channel (monai)
          const float cw = (a.weight != nullptr && Rd != 1) ? a.weight[r] : 1.f;
          gd = dice_scale * cw * dfdp * (smx ? 1.f : p * (1.f - p));
        }
        if (smx) { dldp[r] = gd; dot += gd * p; continue; }
        float gc;
        if (R == 1) {
          const float pw = a.weight ? a.weight[0] : 1.f;
          gc = -pw * yv[r] * (1.f - p) + (1.f - yv[r]) * p;
        } else {
          const float sp = expf(lg[r] - m) / se;
          gc = sp * S - (a.weight ? a.weight[r] : 1.f) * yv[r];
        }
        g[r] = gd + ce_scale * gc;
        a.dz.p[n * a.dz.sn + (long long)r * a.dz.sc + zz * a.dz.sd + y * a.dz.sh + x * a.dz.sw] = g[r];
      }
    if (smx) {
#pragma unroll
      for (int r = 0; r < DCE_MAX_R; ++r)
        if (r < R) {
          const float sp = expf(lg[r] - m) / se;
          const float gc = sp * S - (a.weight ? a.weight[r] : 1.f) * yv[r];
          g[r] = sp * (dldp[r] - dot) + ce_scale * gc;
          a.dz.p[n * a.dz.sn + (long long)r * a.dz.sc + zz * a.dz.sd + y * a.dz.sh + x * a.dz.sw] = g[r];
        }
    }
  }
}

}  // namespace mmtta

using namespace mmtta;

extern "C" int64_t mmtta_entropy_partials(const mmtta_tensor* logits) {
  if (logits == nullptr) return -1;
  return entropy_blocks(logits);
}

static int entropy_launch(const mmtta_tensor* logits, int softmax, const mmtta_tensor* dlogits, double* partial, float* loss,
                          int per_item, void* stream) {
  MMTTA_CHECK(logits == nullptr || logits->dtype == MMTTA_F32, MMTTA_ERR_UNSUPPORTED, "mmtta_entropy_loss: `logits` must be fp32-stored");
  MMTTA_CHECK(logits && dlogits && partial && loss && logits->ptr && dlogits->ptr, MMTTA_ERR_INVALID, "entropy: null argument");
  MMTTA_CHECK(logits->n == dlogits->n && logits->c == dlogits->c && logits->d == dlogits->d && logits->h == dlogits->h &&
                  logits->w == dlogits->w, MMTTA_ERR_INVALID, "entropy: shape mismatch");
  MMTTA_CHECK(is_cl(logits) && is_cl(dlogits), MMTTA_ERR_UNSUPPORTED, "entropy: channels-last only");
  hipStream_t s = (hipStream_t)stream;
  // per_item: every batch item is its own objective - the launch geometry, block partials and scale of ONE item, repeated
  // along gridDim.y (bit-identical to N separate calls)
  mmtta_tensor one = *logits;
  if (per_item) one.n = 1;
  const int items = per_item ? logits->n : 1;
  const int blocks = entropy_blocks(&one);
  const long long nvox = (long long)one.n * logits->d * logits->h * logits->w;
  const dim3 grid(blocks, items);
  if (!softmax) {
    const double cnt = (double)nvox * logits->c;
    auto dense16 = [](const mmtta_tensor* t) {
      return t->sc == 1 && t->sw == 4 && t->sh == (int64_t)t->w * 4 && t->sd == (int64_t)t->h * t->sh && t->sn % 4 == 0 &&
             ((uintptr_t)t->ptr) % 16 == 0;
    };
    const bool vec = logits->c <= 4 && dense16(logits) && dense16(dlogits) && ((dlogits->flags & MMTTA_TENSOR_OWNS_PAD) || dlogits->c == 4);
    MMTTA_CHECK(is_f32(dlogits) || vec, MMTTA_ERR_UNSUPPORTED, "mmtta_entropy_loss: a bf16-stored `dlogits` needs dense 4-channel voxel rows that own their pad");
    if (vec && is_bf16(dlogits))
      hipLaunchKernelGGL(entropy_bernoulli_vec_kernel<true>, grid, dim3(256), 0, s, tv(logits), tv(dlogits), partial, (float)(1.0 / cnt), per_item);
    else if (vec)
      hipLaunchKernelGGL(entropy_bernoulli_vec_kernel<false>, grid, dim3(256), 0, s, tv(logits), tv(dlogits), partial, (float)(1.0 / cnt), per_item);
    else
      hipLaunchKernelGGL(entropy_bernoulli_kernel, grid, dim3(256), 0, s, tv(logits), tv(dlogits), partial, (float)(1.0 / cnt), per_item);
    int st = launch_status("entropy bernoulli");
    if (st) return st;
    hipLaunchKernelGGL(entropy_finish_kernel, dim3(items), dim3(64), 0, s, partial, blocks, 1.0 / cnt, loss);
  } else {
    MMTTA_CHECK(logits->c <= ENT_MAX_R, MMTTA_ERR_UNSUPPORTED, "entropy softmax: more than %d classes", ENT_MAX_R);
    MMTTA_CHECK(is_f32(dlogits), MMTTA_ERR_UNSUPPORTED, "entropy softmax: `dlogits` must be fp32-stored");
    const double cnt = (double)nvox;
    hipLaunchKernelGGL(entropy_categorical_kernel, grid, dim3(256), 0, s, tv(logits), tv(dlogits), partial, (float)(1.0 / cnt), per_item);
    int st = launch_status("entropy categorical");
    if (st) return st;
    hipLaunchKernelGGL(entropy_finish_kernel, dim3(items), dim3(64), 0, s, partial, blocks, 1.0 / cnt, loss);
  }
  return launch_status("entropy finish");
}

extern "C" int mmtta_entropy_loss(const mmtta_tensor* logits, int softmax, const mmtta_tensor* dlogits, double* partial,
                                  float* loss, void* stream) {
  return entropy_launch(logits, softmax, dlogits, partial, loss, 0, stream);
}

extern "C" int64_t mmtta_entropy_partials_items(const mmtta_tensor* logits) {
  if (logits == nullptr) return -1;
  mmtta_tensor one = *logits;
  one.n = 1;
  return (int64_t)entropy_blocks(&one) * logits->n;
}

extern "C" int mmtta_entropy_loss_items(const mmtta_tensor* logits, int softmax, const mmtta_tensor* dlogits, double* partial,
                                        float* loss, void* stream) {
  return entropy_launch(logits, softmax, dlogits, partial, loss, 1, stream);
}

static int optim_launch(int kind, float* p, const float* g, float* m, float* v, int64_t n, int64_t n_decay, const OptimArgs& a,
                        int32_t* step, hipStream_t s, int sets = 1, int64_t set_stride = 0) {
  MMTTA_CHECK(sets >= 1 && set_stride >= 0 && set_stride % 4 == 0 && (sets == 1 || set_stride >= n), MMTTA_ERR_INVALID,
              "optimizer: %d sets with stride %lld (n = %lld)", sets, (long long)set_stride, (long long)n);
  MMTTA_CHECK(p && g && m && step && n >= 0 && n_decay >= 0 && n_decay <= n, MMTTA_ERR_INVALID, "optimizer: bad argument");
  MMTTA_CHECK(kind == MMTTA_OPTIM_SGD || v != nullptr, MMTTA_ERR_INVALID, "optimizer: Adam/AdamW need the second-moment buffer");
  MMTTA_CHECK(kind >= MMTTA_OPTIM_ADAM && kind <= MMTTA_OPTIM_SGD, MMTTA_ERR_INVALID, "optimizer: kind %d", kind);
  if (n == 0) return MMTTA_OK;
  const bool al = ((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) % 16 == 0 && n_decay % 4 == 0;
  MMTTA_CHECK(al, MMTTA_ERR_UNSUPPORTED, "optimizer: buffers must be 16-byte aligned and n_decay a multiple of 4");
  long long blocks = ((n + 3) / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  const dim3 grid((unsigned)blocks, (unsigned)sets), blk(256);
  const long long ss = (long long)set_stride;
  if (kind == MMTTA_OPTIM_ADAM)
    hipLaunchKernelGGL(optim_kernel<0>, grid, blk, 0, s, p, g, m, v, (long long)n, (long long)n_decay, a, step, ss);
  else if (kind == MMTTA_OPTIM_ADAMW)
    hipLaunchKernelGGL(optim_kernel<1>, grid, blk, 0, s, p, g, m, v, (long long)n, (long long)n_decay, a, step, ss);
  else
    hipLaunchKernelGGL(optim_kernel<2>, grid, blk, 0, s, p, g, m, v, (long long)n, (long long)n_decay, a, step, ss);
  int st = launch_status("optimizer");
  if (st) return st;
  hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(1), 0, s, step);
  return launch_status("optimizer step counter");
}

extern "C" int mmtta_adam_step(float* p, const float* g, float* m, float* v, int64_t n, int64_t n_decay, float lr,
                               float beta1, float beta2, float eps, float weight_decay, int32_t* step, void* stream) {
  OptimArgs a{lr, beta1, beta2, eps, weight_decay, 0.f, 0.f, 0};
  return optim_launch(MMTTA_OPTIM_ADAM, p, g, m, v, n, n_decay, a, step, (hipStream_t)stream);
}

extern "C" int mmtta_optim_step(const mmtta_optim_desc* d, float* p, const float* g, float* m, float* v, int64_t n,
                                int64_t n_decay, int32_t* step, void* stream) {
  MMTTA_CHECK(d != nullptr, MMTTA_ERR_INVALID, "optimizer: null desc");
  OptimArgs a{d->lr, d->beta1, d->beta2, d->eps, d->weight_decay, d->momentum, d->dampening, d->nesterov};
  if (d->kind == MMTTA_OPTIM_SGD)
    MMTTA_CHECK(!(d->nesterov && (d->momentum <= 0.f || d->dampening != 0.f)), MMTTA_ERR_INVALID,
                "optimizer: nesterov needs momentum > 0 and zero dampening (torch.optim.SGD raises the same)");
  return optim_launch(d->kind, p, g, m, v, n, n_decay, a, step, (hipStream_t)stream);
}

extern "C" int mmtta_optim_step_sets(const mmtta_optim_desc* d, float* p, const float* g, float* m, float* v, int64_t n,
                                     int64_t n_decay, int sets, int64_t set_stride, int32_t* step, void* stream) {
  MMTTA_CHECK(d != nullptr, MMTTA_ERR_INVALID, "optimizer: null desc");
  OptimArgs a{d->lr, d->beta1, d->beta2, d->eps, d->weight_decay, d->momentum, d->dampening, d->nesterov};
  if (d->kind == MMTTA_OPTIM_SGD)
    MMTTA_CHECK(!(d->nesterov && (d->momentum <= 0.f || d->dampening != 0.f)), MMTTA_ERR_INVALID,
                "optimizer: nesterov needs momentum > 0 and zero dampening (torch.optim.SGD raises the same)");
  return optim_launch(d->kind, p, g, m, v, n, n_decay, a, step, (hipStream_t)stream, sets, set_stride);
}

extern "C" int mmtta_mask_dice_counts(const mmtta_tensor* logits, const mmtta_tensor* label, float threshold, int64_t* counts,
                                      uint8_t* mask, void* stream) {
  MMTTA_CHECK(logits == nullptr || logits->dtype == MMTTA_F32, MMTTA_ERR_UNSUPPORTED, "mmtta_mask_dice_counts: `logits` must be fp32-stored");
  MMTTA_CHECK(label == nullptr || label->dtype == MMTTA_F32, MMTTA_ERR_UNSUPPORTED, "mmtta_mask_dice_counts: `label` must be fp32-stored");
  MMTTA_CHECK(logits && label && counts && logits->ptr && label->ptr, MMTTA_ERR_INVALID, "dice: null argument");
  MMTTA_CHECK(logits->n == label->n && logits->c == label->c && logits->d == label->d && logits->h == label->h &&
                  logits->w == label->w, MMTTA_ERR_INVALID, "dice: logits/label shape mismatch");
  MMTTA_CHECK(logits->c <= 256, MMTTA_ERR_UNSUPPORTED, "dice: more than 256 regions");
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(counts, 0, (size_t)logits->n * logits->c * 3 * sizeof(int64_t), s);
  MMTTA_CHECK(e == hipSuccess, MMTTA_ERR_LAUNCH, "dice: memset failed: %s", hipGetErrorString(e));
  const long long dhw = (long long)logits->d * logits->h * logits->w;
  int rp = 1;
  while (rp < logits->c) rp <<= 1;
  const int nvl = 256 / rp;
  long long bpn = (dhw + (long long)nvl * 16 - 1) / ((long long)nvl * 16);
  if (bpn < 1) bpn = 1;
  if (bpn > 1024) bpn = 1024;
  hipLaunchKernelGGL(dice_counts_kernel, dim3((unsigned)(bpn * logits->n)), dim3(256), 0, s, tv(logits), tv(label), threshold,
                     (unsigned long long*)counts, mask, (int)bpn);
  return launch_status("dice counts");
}

static long long dice_ce_blocks_per_n(const mmtta_tensor* logits) {
  const long long dhw = (long long)logits->d * logits->h * logits->w;
  long long bpn = (dhw + 256 * 4 - 1) / (256 * 4);
  if (bpn < 1) bpn = 1;
  if (bpn > 1024) bpn = 1024;
  return bpn;
}

extern "C" int64_t mmtta_dice_ce_scratch_bytes(const mmtta_tensor* logits) {
  if (logits == nullptr || logits->c < 1 || logits->c > DCE_MAX_R) return -1;
  return (int64_t)logits->n * dice_ce_blocks_per_n(logits) * (logits->c * 3 + 1) * (int64_t)sizeof(double);
}

extern "C" int mmtta_dice_ce_sums(const mmtta_tensor* logits, const mmtta_tensor* label, const float* weight,
                                  int squared_pred, int softmax, double* out, void* scratch, void* stream) {
  MMTTA_CHECK(logits == nullptr || logits->dtype == MMTTA_F32, MMTTA_ERR_UNSUPPORTED, "mmtta_dice_ce_sums: `logits` must be fp32-stored");
  MMTTA_CHECK(label == nullptr || label->dtype == MMTTA_F32, MMTTA_ERR_UNSUPPORTED, "mmtta_dice_ce_sums: `label` must be fp32-stored");
  MMTTA_CHECK(logits && label && out && scratch && logits->ptr && label->ptr, MMTTA_ERR_INVALID, "dice_ce: null argument");
  MMTTA_CHECK(logits->n == label->n && logits->c == label->c && logits->d == label->d && logits->h == label->h &&
                  logits->w == label->w, MMTTA_ERR_INVALID, "dice_ce: logits/label shape mismatch");
  MMTTA_CHECK(logits->c <= DCE_MAX_R, MMTTA_ERR_UNSUPPORTED, "dice_ce: more than %d regions", DCE_MAX_R);
  hipStream_t s = (hipStream_t)stream;
  const long long bpn = dice_ce_blocks_per_n(logits);
  const int cols = logits->c * 3 + 1;
  hipLaunchKernelGGL(dice_ce_sums_kernel, dim3((unsigned)(bpn * logits->n)), dim3(256), 0, s, tv(logits), tv(label), weight,
                     squared_pred, softmax, (double*)scratch, (int)bpn);
  int st = launch_status("dice_ce sums");
  if (st) return st;
  hipLaunchKernelGGL(dice_ce_finish_kernel, dim3(logits->n * cols), dim3(64), 0, s, (const double*)scratch, (int)bpn, cols, out);
  return launch_status("dice_ce finish");
}

extern "C" int mmtta_dice_ce_grad(const mmtta_tensor* logits, const mmtta_tensor* label, const float* weight, int squared_pred,
                                  int softmax, int jaccard, int include_background, float lambda_dice, float lambda_ce, float smooth_nr,
                                  float smooth_dr, const double* sums, const mmtta_tensor* dlogits, void* stream) {
  MMTTA_CHECK(logits == nullptr || logits->dtype == MMTTA_F32, MMTTA_ERR_UNSUPPORTED, "mmtta_dice_ce_grad: `logits` must be fp32-stored");
  MMTTA_CHECK(label == nullptr || label->dtype == MMTTA_F32, MMTTA_ERR_UNSUPPORTED, "mmtta_dice_ce_grad: `label` must be fp32-stored");
  MMTTA_CHECK(dlogits == nullptr || dlogits->dtype == MMTTA_F32, MMTTA_ERR_UNSUPPORTED, "mmtta_dice_ce_grad: `dlogits` must be fp32-stored");
  MMTTA_CHECK(logits && label && dlogits && sums && logits->ptr && label->ptr && dlogits->ptr, MMTTA_ERR_INVALID,
              "dice_ce grad: null argument");
  MMTTA_CHECK(logits->n == label->n && logits->c == label->c && logits->d == label->d && logits->h == label->h &&
                  logits->w == label->w && logits->n == dlogits->n && logits->c == dlogits->c && logits->d == dlogits->d &&
                  logits->h == dlogits->h && logits->w == dlogits->w, MMTTA_ERR_INVALID, "dice_ce grad: shape mismatch");
  MMTTA_CHECK(logits->c <= DCE_MAX_R, MMTTA_ERR_UNSUPPORTED, "dice_ce grad: more than %d regions", DCE_MAX_R);
  DceGradArgs a;
  a.z = tv(logits); a.lab = tv(label); a.dz = tv(dlogits); a.weight = weight; a.sums = sums;
  a.squared = squared_pred; a.jaccard = jaccard; a.include_background = include_background; a.softmax = softmax;
  a.lambda_dice = lambda_dice; a.lambda_ce = lambda_ce; a.smooth_nr = smooth_nr; a.smooth_dr = smooth_dr;
  const long long total = (long long)logits->n * logits->d * logits->h * logits->w;
  long long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(dice_ce_grad_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  return launch_status("dice_ce grad");
}
