// Surface metrics of the evaluation tail: 95th-percentile Hausdorff distance and average surface distance per
// (volume, region), the optional block of the reference evaluator (src/evaluation/seg_eval.py:312-360; SURVEY.md
// section 8f row 4).  The reference calls MONAI (`HausdorffDistanceMetric(percentile=95, directed=False)`,
// `compute_average_surface_distance`), which runs scipy on the host; restated here for the GPU:
//
//   edge(A)   = A & ~erode(A)            6-neighbourhood, outside the volume counts as background
//   d(A->B)   = for every voxel of edge(A): Euclidean distance (spacing-weighted) to the nearest voxel of edge(B)
//   hd        = max(quantile_q(d(P->G)), quantile_q(d(G->P)))      linear interpolation, float32 like torch.quantile
//   asd       = mean(d(P->G))            (mean over both directions when symmetric)
//
// The distance transform is the exact separable one, brute force per axis (no envelope tricks: every candidate is
// tried, so the result is the true minimum; ties cannot change it):
//   pass W   per row: index distance to the nearest edge voxel of the row (int16)
//   pass H   P2(d,h,w) = min_h' ((h-h')*sh)^2 + (g(d,h',w)*sw)^2                fp64, candidates staged in LDS
//   pass D   only at the voxels of the other edge set: min_d' ((d-d')*sd)^2 + P2(d',h,w), then sqrt -> float32
// Distances are appended to a per-(mask, direction) list; the order statistics come from a radix select on the float
// bit patterns and the mean from an integer fixed-point sum, so both are independent of the append order: results are
// bitwise reproducible.  Integer/byte work, HBM/L2-bound; nothing here is GEMM-shaped.
#include "common.h"

namespace mmtta {

constexpr int SURF_INF_G = 32767;
constexpr int SURF_MAX_DIM = 1024;           // int16 row distances, LDS tile H x 64 x 2 B <= 128 KB
constexpr double SURF_FIX = 4294967296.0;    // 2^32 fixed-point scale of the distance sum

struct SurfArgs {
  const unsigned char* pred;   // [M][V]
  TV lab;                      // labels, M = n * c
  int M, D, H, W;
  long long V;
  double sd, sh, sw;
  unsigned char* edges;        // [2][M][V]   0: prediction edges, 1: ground-truth edges
  short* g;                    // [2][M][V]
  double* p2;                  // [2][M][V]
  float* list;                 // [2][M][V]
  unsigned int* cnt;           // [2][M]
  unsigned long long* sum;     // [2][M]
};

// K0: edge voxels of both masks.  All 7 + 7 loads are issued unconditionally from clamped coordinates.
__global__ __launch_bounds__(256) void surf_edges_kernel(SurfArgs a) {
  const int m = blockIdx.y;
  const long long v = (long long)blockIdx.x * 256 + threadIdx.x;
  if (v >= a.V) return;
  long long t = v;
  const int x = (int)(t % a.W); t /= a.W;
  const int y = (int)(t % a.H);
  const int z = (int)(t / a.H);
  const int n = m / a.lab.c, r = m % a.lab.c;
  const unsigned char* pm = a.pred + (long long)m * a.V;
  const float* lp = a.lab.p + (long long)n * a.lab.sn + (long long)r * a.lab.sc;
  const int zs[7] = {z, z > 0 ? z - 1 : z, z + 1 < a.D ? z + 1 : z, z, z, z, z};
  const int ys[7] = {y, y, y, y > 0 ? y - 1 : y, y + 1 < a.H ? y + 1 : y, y, y};
  const int xs[7] = {x, x, x, x, x, x > 0 ? x - 1 : x, x + 1 < a.W ? x + 1 : x};
  const bool in[7] = {true, z > 0, z + 1 < a.D, y > 0, y + 1 < a.H, x > 0, x + 1 < a.W};
  unsigned char pv[7];
  float gv[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    pv[j] = pm[((long long)zs[j] * a.H + ys[j]) * a.W + xs[j]];
    gv[j] = lp[(long long)zs[j] * a.lab.sd + (long long)ys[j] * a.lab.sh + (long long)xs[j] * a.lab.sw];
  }
  bool pin = true, gin = true;     // all six neighbours inside the mask
#pragma unroll
  for (int j = 1; j < 7; ++j) {
    pin = pin && in[j] && pv[j] != 0;
    gin = gin && in[j] && gv[j] > 0.5f;
  }
  a.edges[(long long)m * a.V + v] = (unsigned char)((pv[0] != 0 && !pin) ? 1 : 0);
  a.edges[((long long)a.M + m) * a.V + v] = (unsigned char)((gv[0] > 0.5f && !gin) ? 1 : 0);
}

// K1: per row, index distance to the nearest edge voxel along W (SURF_INF_G when the row has none); also counts
// nothing else: one thread per row, two sweeps.
__global__ __launch_bounds__(256) void surf_scan_w_kernel(SurfArgs a, long long rows) {
  const long long row = (long long)blockIdx.x * 256 + threadIdx.x;
  if (row >= rows) return;
  const unsigned char* e = a.edges + row * a.W;
  short* g = a.g + row * a.W;
  int last = -SURF_INF_G;
  for (int x = 0; x < a.W; ++x) {
    if (e[x]) last = x;
    const int dl = x - last;
    g[x] = (short)(dl < SURF_INF_G ? dl : SURF_INF_G);
  }
  int next = 2 * SURF_INF_G;
  for (int x = a.W - 1; x >= 0; --x) {
    if (e[x]) next = x;
    const int dr = next - x;
    const int cur = g[x];
    g[x] = (short)(dr < cur ? dr : cur);
  }
}

// K2: min-plus along H.  Block = 64 columns (w) x 4 rows; the candidates g(d, 0..H-1, w0..w0+63) sit in LDS.
__global__ __launch_bounds__(256) void surf_pass_h_kernel(SurfArgs a) {
  extern __shared__ short tile[];           // [H][64]
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int w = blockIdx.x * 64 + tx;
  const int d = blockIdx.y;
  const long long sm = blockIdx.z;          // set * M + m
  const long long base = (sm * a.D + d) * (long long)a.H * a.W;
  const int wc = w < a.W ? w : a.W - 1;
  for (int h = ty; h < a.H; h += 4) tile[h * 64 + tx] = a.g[base + (long long)h * a.W + wc];
  __syncthreads();
  if (w >= a.W) return;
  for (int h = ty; h < a.H; h += 4) {
    double best = INFINITY;
    for (int hp = 0; hp < a.H; ++hp) {
      const int gg = tile[hp * 64 + tx];
      const double fw = (double)gg * a.sw;
      const double fh = (double)(h - hp) * a.sh;
      const double val = gg == SURF_INF_G ? (double)INFINITY : fh * fh + fw * fw;
      best = val < best ? val : best;
    }
    a.p2[base + (long long)h * a.W + w] = best;
  }
}

// K3: at the edge voxels of set `dir`, the distance to the other set: min-plus along D over P2 of the other set.
// The loop runs for a whole wave when any lane holds an edge voxel; loads are unconditional (8 in flight).
__global__ __launch_bounds__(256) void surf_pass_d_kernel(SurfArgs a) {
  __shared__ unsigned long long s_sum;
  const int dir = blockIdx.z, m = blockIdx.y;
  const long long v = (long long)blockIdx.x * 256 + threadIdx.x;
  if (threadIdx.x == 0) s_sum = 0ull;
  __syncthreads();
  const long long vc = v < a.V ? v : a.V - 1;
  const long long sa = (long long)dir * a.M + m, sb = (long long)(1 - dir) * a.M + m;
  const bool edge = v < a.V && a.edges[sa * a.V + vc] != 0;
  const unsigned long long bal = __ballot(edge);
  if (bal != 0ull) {
    const long long hw = (long long)a.H * a.W;
    const int z = (int)(vc / hw);
    const long long col = vc - (long long)z * hw;
    const double* p = a.p2 + sb * a.V + col;
    double best = INFINITY;
    int dp = 0;
    for (; dp + 8 <= a.D; dp += 8) {
      double q[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) q[j] = p[(long long)(dp + j) * hw];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const double fd = (double)(z - dp - j) * a.sd;
        const double val = fd * fd + q[j];
        best = val < best ? val : best;
      }
    }
    for (; dp < a.D; ++dp) {
      const double fd = (double)(z - dp) * a.sd;
      const double val = fd * fd + p[(long long)dp * hw];
      best = val < best ? val : best;
    }
    const float dist = (float)sqrt(best);
    const int lane = threadIdx.x & 63;
    const int first = __ffsll((long long)bal) - 1;
    unsigned int basei = 0;
    if (lane == first) basei = atomicAdd(a.cnt + sa, (unsigned int)__popcll(bal));
    basei = __shfl(basei, first);
    if (edge) {
      const unsigned int idx = basei + (unsigned int)__popcll(bal & ((1ull << lane) - 1ull));
      a.list[sa * a.V + idx] = dist;
      if (dist < INFINITY) atomicAdd(&s_sum, (unsigned long long)((double)dist * SURF_FIX));
    }
  }
  __syncthreads();
  if (threadIdx.x == 0 && s_sum != 0ull) atomicAdd(a.sum + sa, s_sum);
}

// K4: per mask: the two directional quantiles (radix select, 8 bits per round) -> hd; fixed-point sums -> asd.
__device__ float surf_quantile(const float* list, unsigned int n, float q, unsigned int* hist, unsigned int* sh) {
  // torch.quantile(linear) in float32: rank = q * (n - 1); lerp(v[floor], v[ceil], rank - floor)
  const float rank = q * (float)(n - 1);
  const unsigned int lo = (unsigned int)rank;
  const float wgt = rank - (float)lo;
  const bool need_hi = ceilf(rank) != (float)lo;
  unsigned int prefix = 0, maskbits = 0, k = lo;
  for (int pass = 3; pass >= 0; --pass) {
    for (int i = threadIdx.x; i < 256; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    for (unsigned int i = threadIdx.x; i < n; i += blockDim.x) {
      const unsigned int b = __float_as_uint(list[i]);
      if ((b & maskbits) == prefix) atomicAdd(&hist[(b >> (8 * pass)) & 255u], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned int cum = 0, bsel = 255;
      for (unsigned int bkt = 0; bkt < 256; ++bkt) {
        if (cum + hist[bkt] > k) { bsel = bkt; break; }
        cum += hist[bkt];
      }
      sh[0] = bsel; sh[1] = k - cum;
    }
    __syncthreads();
    prefix |= sh[0] << (8 * pass);
    k = sh[1];
    maskbits |= 255u << (8 * pass);
    __syncthreads();
  }
  const float vlo = __uint_as_float(prefix);
  if (!need_hi) return vlo;
  // v[lo + 1]: vlo again when enough copies of it exist, else the smallest larger value
  if (threadIdx.x == 0) { sh[0] = 0; sh[1] = 0xffffffffu; }
  __syncthreads();
  unsigned int cle = 0, nmin = 0xffffffffu;
  for (unsigned int i = threadIdx.x; i < n; i += blockDim.x) {
    const unsigned int b = __float_as_uint(list[i]);
    if (b <= prefix) ++cle;
    else nmin = b < nmin ? b : nmin;
  }
  atomicAdd(&sh[0], cle);
  atomicMin(&sh[1], nmin);
  __syncthreads();
  const float vhi = (lo + 1 < sh[0]) ? vlo : __uint_as_float(sh[1]);
  __syncthreads();
  const float diff = vhi - vlo;
  return wgt < 0.5f ? fmaf(wgt, diff, vlo) : fmaf(wgt - 1.0f, diff, vhi);
}

__global__ __launch_bounds__(1024) void surf_finish_kernel(SurfArgs a, float q, int use_max, int symmetric, float* hd,
                                                           float* asd) {
  __shared__ unsigned int hist[256];
  __shared__ unsigned int sh[2];
  const int m = blockIdx.x;
  const unsigned int np = a.cnt[m], ng = a.cnt[a.M + m];
  float hdv, asdv;
  if (np == 0 || ng == 0) {
    // MONAI: no edges on one side -> all distances are inf (quantile of infs is nan); none at all -> nan
    hdv = NAN;
    asdv = (np == 0 && ng == 0) ? NAN : INFINITY;
  } else {
    const float qq = use_max ? 1.0f : q;
    const float q0 = surf_quantile(a.list + (long long)m * a.V, np, qq, hist, sh);
    const float q1 = surf_quantile(a.list + ((long long)a.M + m) * a.V, ng, qq, hist, sh);
    hdv = q0 > q1 ? q0 : q1;
    double s = (double)a.sum[m];
    double cnt = (double)np;
    if (symmetric) { s += (double)a.sum[a.M + m]; cnt += (double)ng; }
    asdv = (float)(s / SURF_FIX / cnt);
  }
  if (threadIdx.x == 0) { hd[m] = hdv; asd[m] = asdv; }
}

static size_t surf_align(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace mmtta

using namespace mmtta;

extern "C" int64_t mmtta_surface_scratch_bytes(int64_t n_masks, int64_t d, int64_t h, int64_t w) {
  if (n_masks < 1 || d < 1 || h < 1 || w < 1 || d > SURF_MAX_DIM || h > SURF_MAX_DIM || w > SURF_MAX_DIM) return -1;
  const size_t mv = (size_t)2 * n_masks * d * h * w;
  return (int64_t)(surf_align((size_t)2 * n_masks * 16) + surf_align(mv * 8) + surf_align(mv * 4) + surf_align(mv * 2) +
                   surf_align(mv));
}

extern "C" int mmtta_surface_distances(const uint8_t* pred_mask, const mmtta_tensor* label, const double* spacing,
                                       double percentile, int asd_symmetric, float* hd, float* asd, void* scratch,
                                       void* stream) {
  MMTTA_CHECK(label == nullptr || label->dtype == MMTTA_F32, MMTTA_ERR_UNSUPPORTED, "mmtta_surface_distances: `label` must be fp32-stored");
  MMTTA_CHECK(pred_mask && label && label->ptr && spacing && hd && asd && scratch, MMTTA_ERR_INVALID, "surface: null argument");
  MMTTA_CHECK(label->n >= 1 && label->c >= 1 && label->d >= 1 && label->h >= 1 && label->w >= 1, MMTTA_ERR_INVALID,
              "surface: empty label tensor");
  MMTTA_CHECK(label->d <= SURF_MAX_DIM && label->h <= SURF_MAX_DIM && label->w <= SURF_MAX_DIM, MMTTA_ERR_UNSUPPORTED,
              "surface: spatial extent above %d", SURF_MAX_DIM);
  MMTTA_CHECK(percentile >= 0.0 && percentile <= 100.0, MMTTA_ERR_INVALID, "surface: percentile should be within [0, 100], got %g",
              percentile);
  MMTTA_CHECK(spacing[0] > 0.0 && spacing[1] > 0.0 && spacing[2] > 0.0, MMTTA_ERR_INVALID, "surface: spacing must be positive");
  const long long M = (long long)label->n * label->c;
  MMTTA_CHECK(M <= 65535, MMTTA_ERR_UNSUPPORTED, "surface: more than 65535 masks per call");
  hipStream_t s = (hipStream_t)stream;
  SurfArgs a;
  a.pred = pred_mask;
  a.lab = tv(label);
  a.M = (int)M; a.D = label->d; a.H = label->h; a.W = label->w;
  a.V = (long long)a.D * a.H * a.W;
  a.sd = spacing[0]; a.sh = spacing[1]; a.sw = spacing[2];
  const size_t mv = (size_t)2 * M * a.V;
  char* base = (char*)scratch;
  const size_t head = surf_align((size_t)2 * M * 16);
  a.sum = (unsigned long long*)base;
  a.cnt = (unsigned int*)(base + (size_t)2 * M * 8);
  a.p2 = (double*)(base + head);
  a.list = (float*)(base + head + surf_align(mv * 8));
  a.g = (short*)(base + head + surf_align(mv * 8) + surf_align(mv * 4));
  a.edges = (unsigned char*)(base + head + surf_align(mv * 8) + surf_align(mv * 4) + surf_align(mv * 2));
  hipError_t e = hipMemsetAsync(base, 0, head, s);
  MMTTA_CHECK(e == hipSuccess, MMTTA_ERR_LAUNCH, "surface: memset failed: %s", hipGetErrorString(e));
  const unsigned vb = (unsigned)((a.V + 255) / 256);
  hipLaunchKernelGGL(surf_edges_kernel, dim3(vb, (unsigned)M), dim3(256), 0, s, a);
  int st = launch_status("surface edges");
  if (st) return st;
  const long long rows = 2 * M * a.D * a.H;
  hipLaunchKernelGGL(surf_scan_w_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, a, rows);
  st = launch_status("surface scan");
  if (st) return st;
  const size_t lds = (size_t)a.H * 64 * sizeof(short);
  if (lds > 48 * 1024) {
    e = hipFuncSetAttribute((const void*)surf_pass_h_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    MMTTA_CHECK(e == hipSuccess, MMTTA_ERR_LAUNCH, "surface: LDS size %zu refused: %s", lds, hipGetErrorString(e));
  }
  hipLaunchKernelGGL(surf_pass_h_kernel, dim3((unsigned)((a.W + 63) / 64), (unsigned)a.D, (unsigned)(2 * M)), dim3(256), lds, s, a);
  st = launch_status("surface pass h");
  if (st) return st;
  hipLaunchKernelGGL(surf_pass_d_kernel, dim3(vb, (unsigned)M, 2), dim3(256), 0, s, a);
  st = launch_status("surface pass d");
  if (st) return st;
  hipLaunchKernelGGL(surf_finish_kernel, dim3((unsigned)M), dim3(1024), 0, s, a, (float)(percentile / 100.0),
                     percentile >= 100.0 ? 1 : 0, asd_symmetric ? 1 : 0, hd, asd);
  return launch_status("surface finish");
}
