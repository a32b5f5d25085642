// Input normalisation pre-pass (the step immediately before the hot path).
//
// Restates `_normalize_img` of the reference (src/datasets/transforms.py:129-223) for one image [C,D,H,W]:
//   (A) intensity policy, per channel: optional clip [lo,hi]; optional z-score with statistics over the voxels
//       x > mask_gt of the CLIPPED channel (all voxels when fewer than min_count qualify, or when not masked):
//       mu = mean, sd = max(std(unbiased=False), eps), x = (x - mu) / sd            (:163-198)
//   (B) legacy per-channel (x - mean) / std                                          (:202-223)
// HBM-bound: one read for the statistics (fp64 partial sums per block, fixed order: deterministic), one read +
// one write for the apply.  4 B * (2 reads + 1 write) per voxel and channel.
#include "common.h"

namespace mmtta {

constexpr int PRE_MAXC = 8;
constexpr int PRE_BLOCKS = 256;     // partial rows per channel

struct PreArgs {
  const float* x; float* y;
  long long xsc, ysc;               // channel strides (elements); voxels of a channel are contiguous
  long long dhw;
  int C;
  mmtta_intensity_rule rule[PRE_MAXC];
  double* part;                     // [C][PRE_BLOCKS][5]: n_masked, sum_masked, sumsq_masked, sum_all, sumsq_all
  float* coef;                      // [C][2]: mu, sd
};

__global__ __launch_bounds__(256) void intensity_stats_kernel(PreArgs a) {
  __shared__ double sh[5][4];
  const int c = blockIdx.y;
  const mmtta_intensity_rule r = a.rule[c];
  const float* xp = a.x + (long long)c * a.xsc;
  double acc[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < a.dhw; i += (long long)gridDim.x * 256) {
    float v = xp[i];
    if (r.clip) v = fminf(fmaxf(v, r.lo), r.hi);
    const double d = (double)v;
    if (v > r.mask_gt) { acc[0] += 1.0; acc[1] += d; acc[2] += d * d; }
    acc[3] += d; acc[4] += d * d;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < 5; ++q) {
    const double s = wave_sum_d(acc[q]);
    if (lane == 0) sh[q][wave] = s;
  }
  __syncthreads();
  if (threadIdx.x < 5)
    a.part[((long long)c * PRE_BLOCKS + blockIdx.x) * 5 + threadIdx.x] =
        (sh[threadIdx.x][0] + sh[threadIdx.x][1]) + (sh[threadIdx.x][2] + sh[threadIdx.x][3]);
}

__global__ __launch_bounds__(64) void intensity_finalize_kernel(PreArgs a, int nblocks) {
  const int c = blockIdx.x;
  const mmtta_intensity_rule r = a.rule[c];
  double t[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  for (int b = threadIdx.x; b < nblocks; b += 64)
#pragma unroll
    for (int q = 0; q < 5; ++q) t[q] += a.part[((long long)c * PRE_BLOCKS + b) * 5 + q];
#pragma unroll
  for (int q = 0; q < 5; ++q) t[q] = wave_sum_d(t[q]);
  if (threadIdx.x == 0) {
    double n, s, ss;
    if (r.masked && t[0] >= (double)r.min_count) { n = t[0]; s = t[1]; ss = t[2]; }
    else { n = (double)a.dhw; s = t[3]; ss = t[4]; }
    const double mu = s / n;
    double var = ss / n - mu * mu;
    if (var < 0.0) var = 0.0;
    float sd = (float)sqrt(var);
    if (sd < r.eps) sd = r.eps;
    a.coef[2 * c] = (float)mu;
    a.coef[2 * c + 1] = sd;
  }
}

__global__ __launch_bounds__(256) void intensity_apply_kernel(PreArgs a) {
  const int c = blockIdx.y;
  const mmtta_intensity_rule r = a.rule[c];
  const float* xp = a.x + (long long)c * a.xsc;
  float* yp = a.y + (long long)c * a.ysc;
  float mu = 0.f, sd = 1.f;
  if (r.legacy) { mu = r.mean; sd = r.std; }
  else if (r.zscore) { mu = a.coef[2 * c]; sd = a.coef[2 * c + 1]; }
  const bool scale = r.legacy || r.zscore;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < a.dhw; i += (long long)gridDim.x * 256) {
    float v = xp[i];
    if (!r.legacy && r.clip) v = fminf(fmaxf(v, r.lo), r.hi);
    yp[i] = scale ? (v - mu) / sd : v;
  }
}

}  // namespace mmtta

using namespace mmtta;

extern "C" int64_t mmtta_intensity_scratch_bytes(int channels) {
  if (channels < 1 || channels > PRE_MAXC) return -1;
  return (int64_t)channels * PRE_BLOCKS * 5 * (int64_t)sizeof(double) + (int64_t)channels * 2 * (int64_t)sizeof(float);
}

extern "C" int mmtta_intensity_normalize(const mmtta_tensor* x, const mmtta_intensity_rule* rules, const mmtta_tensor* y,
                                         void* scratch, void* stream) {
  MMTTA_CHECK(x == nullptr || x->dtype == MMTTA_F32, MMTTA_ERR_UNSUPPORTED, "mmtta_intensity_normalize: `x` must be fp32-stored");
  MMTTA_CHECK(x && y && rules && scratch && x->ptr && y->ptr, MMTTA_ERR_INVALID, "intensity: null argument");
  MMTTA_CHECK(x->n == 1 && y->n == 1, MMTTA_ERR_UNSUPPORTED, "intensity: one image per call (statistics are per image)");
  MMTTA_CHECK(x->c == y->c && x->d == y->d && x->h == y->h && x->w == y->w, MMTTA_ERR_INVALID, "intensity: shape mismatch");
  MMTTA_CHECK(x->c >= 1 && x->c <= PRE_MAXC, MMTTA_ERR_UNSUPPORTED, "intensity: %d channels (max %d)", x->c, PRE_MAXC);
  auto dense = [](const mmtta_tensor* t) { return t->sw == 1 && t->sh == t->w && t->sd == (int64_t)t->h * t->w; };
  MMTTA_CHECK(dense(x) && dense(y), MMTTA_ERR_UNSUPPORTED, "intensity: tensors must be [C,D,H,W] with dense voxels (NCDHW boundary layout)");
  PreArgs a;
  a.x = (const float*)x->ptr; a.y = (float*)y->ptr; a.xsc = x->sc; a.ysc = y->sc;
  a.dhw = (long long)x->d * x->h * x->w; a.C = x->c;
  bool need_stats = false;
  for (int c = 0; c < x->c; ++c) {
    a.rule[c] = rules[c];
    if (!rules[c].legacy && rules[c].zscore) need_stats = true;
    if (rules[c].legacy) MMTTA_CHECK(rules[c].std != 0.f, MMTTA_ERR_INVALID, "intensity: std of channel %d is 0", c);
  }
  for (int c = x->c; c < PRE_MAXC; ++c) a.rule[c] = rules[0];
  a.part = (double*)scratch;
  a.coef = (float*)((char*)scratch + (size_t)x->c * PRE_BLOCKS * 5 * sizeof(double));
  hipStream_t s = (hipStream_t)stream;
  long long blocks = (a.dhw + 255) / 256;
  const int sblocks = (int)(blocks < PRE_BLOCKS ? blocks : PRE_BLOCKS);
  if (need_stats) {
    hipLaunchKernelGGL(intensity_stats_kernel, dim3(sblocks, x->c), dim3(256), 0, s, a);
    int st = launch_status("intensity stats");
    if (st) return st;
    hipLaunchKernelGGL(intensity_finalize_kernel, dim3(x->c), dim3(64), 0, s, a, sblocks);
    st = launch_status("intensity finalize");
    if (st) return st;
  }
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(intensity_apply_kernel, dim3((unsigned)blocks, x->c), dim3(256), 0, s, a);
  return launch_status("intensity apply");
}
