// HBM-bound kernels of the adaptation path (gfx950): layout copies, norm statistics and the
// norm(+ReLU) backward, the residual "combine", trilinear x2 and its adjoint, linear
// combinations (modality means).  All tensors are channels-last unless noted; channel is the
// fastest thread index so a wave touches 256 contiguous bytes (4 B/lane) or 1 KiB (16 B/lane,
// the float4 paths) per instruction.  Reductions are two-stage and deterministic: partial slabs
// [rows][2][C] in fp32, combined in fp64 - no float atomics anywhere.
#include "common.h"

namespace mmtta {


// ------------------------------------------------------------------ strided copy
template <bool DBF>
__global__ void copy_strided_kernel(TV s, TV d) {
  const long long total = (long long)d.n * d.c * d.d * d.h * d.w;
  // iterate in the destination's fastest order: if dst is channels-last, c fastest; else x fastest
  const bool cl = d.sc == 1;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    int n, c, z, y, x;
    long long t = i;
    if (cl) {
      c = (int)(t % d.c); t /= d.c;
      x = (int)(t % d.w); t /= d.w;
      y = (int)(t % d.h); t /= d.h;
      z = (int)(t % d.d); n = (int)(t / d.d);
    } else {
      x = (int)(t % d.w); t /= d.w;
      y = (int)(t % d.h); t /= d.h;
      z = (int)(t % d.d); t /= d.d;
      c = (int)(t % d.c); n = (int)(t / d.c);
    }
    st1_any(d.p, vox_addr(d, n, z, y, x) + (long long)c * d.sc, s.p[vox_addr(s, n, z, y, x) + (long long)c * s.sc], DBF);
  }
}

// ------------------------------------------------------------------ two-stage channel reductions
// Block layout: cpl channel lanes (power of two >= min(C,256)) x (256/cpl) voxel lanes.
// MODE 0: (sum x, sum x^2).  MODE 1: norm backward (sum dz, sum dz*xhat).
struct RedArgs {
  TV x;        // MODE 0: tensor ; MODE 1: y (pre-norm)
  TV dout;     // MODE 1
  NL t;        // MODE 1
  float* part; // [N*rows_per_n][2][C]
  int rows_per_n;
  long long vox_per_row;
};

// VEC = 4: every thread owns 4 consecutive channels (16-byte loads); VEC = 1: scalar fallback.
// `regular` tensors (dense voxel order: sh == W*sw, sd == H*sh) are addressed as voxel*sw, no div/mod.
// XBF / DBF: storage of x (the activation) and of dout (the gradient: bf16 when method.grad_storage stores gradients so)
template <int MODE, int VEC, bool XBF = false, bool DBF = false>
__global__ __launch_bounds__(256) void channel_reduce_kernel(RedArgs a) {
  __shared__ float red[2][VEC][256];
  const int C = a.x.c;
  const int CV = (C + VEC - 1) / VEC;          // channel vectors
  int cpl = 1;
  while (cpl < CV && cpl < 256) cpl <<= 1;
  const int nvl = 256 / cpl;
  const int cl = threadIdx.x % cpl, vl = threadIdx.x / cpl;
  const int n = blockIdx.x / a.rows_per_n, row = blockIdx.x % a.rows_per_n;
  const long long dhw = (long long)a.x.d * a.x.h * a.x.w;
  const long long v0 = row * a.vox_per_row;
  const long long v1 = (v0 + a.vox_per_row < dhw) ? v0 + a.vox_per_row : dhw;
  const bool regx = a.x.sh == (long long)a.x.w * a.x.sw && a.x.sd == (long long)a.x.h * a.x.sh;
  const bool regd = a.dout.sh == (long long)a.dout.w * a.dout.sw && a.dout.sd == (long long)a.dout.h * a.dout.sh;
  for (int cb = 0; cb < CV; cb += cpl) {
    const int c0 = (cb + cl) * VEC;
    float s0[VEC], s1[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { s0[j] = 0.f; s1[j] = 0.f; }
    if (c0 < C) {
      float mu[VEC], rs[VEC], g[VEC], b[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        mu[j] = 0.f; rs[j] = 1.f; g[j] = 1.f; b[j] = 0.f;
        if (MODE == 1 && a.t.mean && c0 + j < C) {
          mu[j] = a.t.mean[n * C + c0 + j]; rs[j] = a.t.rstd[n * C + c0 + j];
          if (a.t.gamma) g[j] = a.t.gamma[c0 + j];
          if (a.t.beta) b[j] = a.t.beta[c0 + j];
        }
      }
      for (long long v = v0 + vl; v < v1; v += nvl) {
        long long ax, ad;
        if (regx && regd) {
          ax = (long long)n * a.x.sn + v * a.x.sw;
          ad = (long long)n * a.dout.sn + v * a.dout.sw;
        } else {
          long long t = v;
          const int xx = (int)(t % a.x.w); t /= a.x.w;
          const int yy = (int)(t % a.x.h);
          const int zz = (int)(t / a.x.h);
          ax = vox_addr(a.x, n, zz, yy, xx);
          ad = vox_addr(a.dout, n, zz, yy, xx);
        }
        float xv[VEC], dv[VEC];
        if (VEC == 4) {
          const float4 t4 = ld4_t<XBF>(a.x.p, ax + c0);               // the pre-norm activation may be bf16-stored
          xv[0] = t4.x; xv[1] = t4.y; xv[2] = t4.z; xv[3] = t4.w;
          if (MODE == 1) {
            const float4 d4 = ld4_t<DBF>(a.dout.p, ad + c0);
            dv[0] = d4.x; dv[1] = d4.y; dv[2] = d4.z; dv[3] = d4.w;
          }
        } else {
          xv[0] = ld1_t<XBF>(a.x.p, ax + c0);
          if (MODE == 1) dv[0] = ld1_t<DBF>(a.dout.p, ad + c0);
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          if (MODE == 0) {
            s0[j] += xv[j];
            s1[j] += xv[j] * xv[j];
          } else {
            const float xhat = (xv[j] - mu[j]) * rs[j];
            float dz = dv[j];
            if (a.t.relu && !(fmaf(g[j], xhat, b[j]) > 0.f)) dz = 0.f;
            s0[j] += dz;
            s1[j] += dz * xhat;
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) { red[0][j][threadIdx.x] = s0[j]; red[1][j][threadIdx.x] = s1[j]; }
    __syncthreads();
    if (vl == 0 && c0 < C) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        if (c0 + j < C) {
          float t0 = 0.f, t1 = 0.f;
          for (int q = 0; q < nvl; ++q) { t0 += red[0][j][q * cpl + cl]; t1 += red[1][j][q * cpl + cl]; }
          a.part[((long long)blockIdx.x * 2 + 0) * C + c0 + j] = t0;
          a.part[((long long)blockIdx.x * 2 + 1) * C + c0 + j] = t1;
        }
      }
    }
    __syncthreads();
  }
}

// stage 2a: one wave per (n,c): fp64 sum over the partial rows -> tot[n*C+c][2]
__global__ __launch_bounds__(64) void rows_reduce_kernel(const float* part, int rows_per_n, int C, double* tot) {
  const int n = blockIdx.x / C, c = blockIdx.x % C;
  double s0 = 0.0, s1 = 0.0;
  for (int r = threadIdx.x; r < rows_per_n; r += 64) {
    const long long row = (long long)n * rows_per_n + r;
    s0 += (double)part[(row * 2 + 0) * C + c];
    s1 += (double)part[(row * 2 + 1) * C + c];
  }
  s0 = wave_sum_d(s0);
  s1 = wave_sum_d(s1);
  if (threadIdx.x == 0) {
    tot[((long long)n * C + c) * 2 + 0] = s0;
    tot[((long long)n * C + c) * 2 + 1] = s1;
  }
}

// InstanceNorm fast path: one wave per (n,c) reduces the partial rows AND finishes the statistics
// (no cross-(n,c) coupling), one launch instead of two.
__global__ __launch_bounds__(64) void instance_stats_kernel(const float* part, int rows_per_n, int C, double count, float eps,
                                                            float* mean, float* rstd, const float* gamma, const float* beta,
                                                            float* scale, float* shift) {
  const int n = blockIdx.x / C, c = blockIdx.x % C;
  double s0 = 0.0, s1 = 0.0;
  for (int r = threadIdx.x; r < rows_per_n; r += 64) {
    const long long row = (long long)n * rows_per_n + r;
    s0 += (double)part[(row * 2 + 0) * C + c];
    s1 += (double)part[(row * 2 + 1) * C + c];
  }
  s0 = wave_sum_d(s0);
  s1 = wave_sum_d(s1);
  if (threadIdx.x == 0) {
    const double mu = s0 / count;
    double var = s1 / count - mu * mu;
    if (var < 0.0) var = 0.0;
    const float rs = (float)(1.0 / sqrt(var + (double)eps));
    mean[blockIdx.x] = (float)mu;
    rstd[blockIdx.x] = rs;
    if (scale != nullptr) {
      const float sc = rs * (gamma ? gamma[c] : 1.f);
      scale[blockIdx.x] = sc;
      shift[blockIdx.x] = (beta ? beta[c] : 0.f) - (float)mu * sc;
    }
  }
}

__global__ __launch_bounds__(64) void instance_bwd_kernel(const float* part, int rows_per_n, int C, double count,
                                                          const float* gamma, float* m1, float* m2) {
  const int n = blockIdx.x / C, c = blockIdx.x % C;
  double s0 = 0.0, s1 = 0.0;
  for (int r = threadIdx.x; r < rows_per_n; r += 64) {
    const long long row = (long long)n * rows_per_n + r;
    s0 += (double)part[(row * 2 + 0) * C + c];
    s1 += (double)part[(row * 2 + 1) * C + c];
  }
  s0 = wave_sum_d(s0);
  s1 = wave_sum_d(s1);
  if (threadIdx.x == 0) {
    const double g = gamma ? (double)gamma[c] : 1.0;
    m1[blockIdx.x] = (float)(g * s0 / count);
    m2[blockIdx.x] = (float)(g * s1 / count);
  }
}

struct StatFinArgs {
  int kind, groups, N, C;
  double count;
  float eps;
  int training;
  float* running_mean; float* running_var; float momentum;
  float* mean; float* rstd;
  const float* gamma; const float* beta; float* scale; float* shift;
  const double* tot;
};

__global__ void stats_finalize_kernel(StatFinArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.N * a.C) return;
  const int n = i / a.C, c = i % a.C;
  double S = 0.0, Q = 0.0, cnt = 0.0;
  if (a.kind == MMTTA_NORM_INSTANCE) {
    S = a.tot[(long long)i * 2]; Q = a.tot[(long long)i * 2 + 1]; cnt = a.count;
  } else if (a.kind == MMTTA_NORM_BATCH) {
    if (!a.training) {
      a.mean[i] = a.running_mean[c];
      a.rstd[i] = (float)(1.0 / sqrt((double)a.running_var[c] + (double)a.eps));
      if (a.scale != nullptr) {
        const float sc = a.rstd[i] * (a.gamma ? a.gamma[c] : 1.f);
        a.scale[i] = sc;
        a.shift[i] = (a.beta ? a.beta[c] : 0.f) - a.mean[i] * sc;
      }
      return;
    }
    for (int m = 0; m < a.N; ++m) { S += a.tot[((long long)m * a.C + c) * 2]; Q += a.tot[((long long)m * a.C + c) * 2 + 1]; }
    cnt = a.count * a.N;
  } else {
    const int cg = a.C / a.groups, g0 = (c / cg) * cg;
    for (int k = g0; k < g0 + cg; ++k) { S += a.tot[((long long)n * a.C + k) * 2]; Q += a.tot[((long long)n * a.C + k) * 2 + 1]; }
    cnt = a.count * cg;
  }
  const double mu = S / cnt;
  double var = Q / cnt - mu * mu;
  if (var < 0.0) var = 0.0;
  a.mean[i] = (float)mu;
  a.rstd[i] = (float)(1.0 / sqrt(var + (double)a.eps));
  if (a.scale != nullptr) {
    const float sc = a.rstd[i] * (a.gamma ? a.gamma[c] : 1.f);
    a.scale[i] = sc;
    a.shift[i] = (a.beta ? a.beta[c] : 0.f) - a.mean[i] * sc;
  }
  if (a.kind == MMTTA_NORM_BATCH && a.training && n == 0 && a.running_mean != nullptr) {
    const double unb = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
    a.running_mean[c] = (float)((1.0 - a.momentum) * a.running_mean[c] + a.momentum * mu);
    a.running_var[c] = (float)((1.0 - a.momentum) * a.running_var[c] + a.momentum * unb);
  }
}

struct BwdFinArgs {
  int kind, groups, N, C;
  double count;
  const float* gamma;
  int training;
  float* m1; float* m2; float* dgamma; float* dbeta; int accumulate;
  const double* tot;
};

__global__ void bwd_finalize_kernel(BwdFinArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.N * a.C) return;
  const int n = i / a.C, c = i % a.C;
  double A = 0.0, B = 0.0, cnt = 1.0;
  if (a.kind == MMTTA_NORM_INSTANCE) {
    const double g = a.gamma ? (double)a.gamma[c] : 1.0;
    A = g * a.tot[(long long)i * 2]; B = g * a.tot[(long long)i * 2 + 1]; cnt = a.count;
  } else if (a.kind == MMTTA_NORM_BATCH) {
    const double g = a.gamma ? (double)a.gamma[c] : 1.0;
    for (int m = 0; m < a.N; ++m) { A += g * a.tot[((long long)m * a.C + c) * 2]; B += g * a.tot[((long long)m * a.C + c) * 2 + 1]; }
    cnt = a.count * a.N;
  } else {
    const int cg = a.C / a.groups, g0 = (c / cg) * cg;
    for (int k = g0; k < g0 + cg; ++k) {
      const double g = a.gamma ? (double)a.gamma[k] : 1.0;
      A += g * a.tot[((long long)n * a.C + k) * 2]; B += g * a.tot[((long long)n * a.C + k) * 2 + 1];
    }
    cnt = a.count * cg;
  }
  const bool frozen = (a.kind == MMTTA_NORM_BATCH && !a.training);
  a.m1[i] = frozen ? 0.f : (float)(A / cnt);
  a.m2[i] = frozen ? 0.f : (float)(B / cnt);
  if (n == 0 && a.dgamma != nullptr) {
    double dg = 0.0, db = 0.0;
    for (int m = 0; m < a.N; ++m) { db += a.tot[((long long)m * a.C + c) * 2]; dg += a.tot[((long long)m * a.C + c) * 2 + 1]; }
    a.dgamma[c] = a.accumulate ? a.dgamma[c] + (float)dg : (float)dg;
    if (a.dbeta) a.dbeta[c] = a.accumulate ? a.dbeta[c] + (float)db : (float)db;
  }
}

// ------------------------------------------------------------------ elementwise (channel fastest)
struct EwArgs {
  TV a, b, o;
  NL ta, tb;
  const float* m1; const float* m2;
  int hasb;
};

// MODE 0: combine  out = Ta(a) + Tb(b)
// MODE 1: norm bwd apply: a = dout, b = y, T = ta(on y)   out = rstd*(g*dz - m1 - xhat*m2)
template <int MODE, int VEC, bool ABF = false, bool BBF = false, bool OBF = false>
__global__ __launch_bounds__(256) void elementwise_kernel(EwArgs e) {
  const int C = e.o.c;
  const int CV = (C + VEC - 1) / VEC;
  const long long nvox = (long long)e.o.n * e.o.d * e.o.h * e.o.w;
  const long long total = nvox * CV;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c0 = (int)(i % CV) * VEC;
    const long long v = i / CV;
    int n, z, y, x;
    vox_decompose(e.o, v, n, z, y, x);
    float av[VEC], bv[VEC], ov[VEC];
    const long long ao = vox_addr(e.a, n, z, y, x) + c0;
    const long long bo = e.hasb ? vox_addr(e.b, n, z, y, x) + c0 : 0;
    const bool bp = e.hasb != 0;
    if (VEC == 4) {
      const float4 t = ld4_t<ABF>(e.a.p, ao);
      av[0] = t.x; av[1] = t.y; av[2] = t.z; av[3] = t.w;
      if (bp) { const float4 u = ld4_t<BBF>(e.b.p, bo); bv[0] = u.x; bv[1] = u.y; bv[2] = u.z; bv[3] = u.w; }
    } else {
      av[0] = ld1_t<ABF>(e.a.p, ao);
      if (bp) bv[0] = ld1_t<BBF>(e.b.p, bo);
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const int c = c0 + j;
      if (MODE == 0) {
        float sa, ha;
        nl_coeff(e.ta, n, C, min(c, C - 1), sa, ha);
        float r = nl_apply(av[j], sa, ha, e.ta.relu);
        if (bp) {
          float sb, hb;
          nl_coeff(e.tb, n, C, min(c, C - 1), sb, hb);
          r += nl_apply(bv[j], sb, hb, e.tb.relu);
        }
        ov[j] = r;
      } else {
        const int cc = min(c, C - 1);                 // pad lanes of a padded row compute on channel C-1's values
        const float mu = e.ta.mean[n * C + cc], rs = e.ta.rstd[n * C + cc];
        const float g = e.ta.gamma ? e.ta.gamma[cc] : 1.f;
        const float bt = e.ta.beta ? e.ta.beta[cc] : 0.f;
        const float xhat = (bv[j] - mu) * rs;
        float dz = av[j];
        if (e.ta.relu && !(fmaf(g, xhat, bt) > 0.f)) dz = 0.f;
        ov[j] = rs * (g * dz - e.m1[n * C + cc] - xhat * e.m2[n * C + cc]);
      }
    }
    const long long oo = vox_addr(e.o, n, z, y, x) + c0;
    if (VEC == 4) st4_t<OBF>(e.o.p, oo, make_float4(ov[0], ov[1], ov[2], ov[3]));
    else st1_t<OBF>(e.o.p, oo, ov[0]);
  }
}

// ------------------------------------------------------------------ norm backward apply, 8 channels per thread
// The same arithmetic as elementwise_kernel<1, ...> for voxel-dense tensors with C a power of two >= 8: a thread keeps ONE
// channel octet (its 48 coefficients are loaded once, as 16-byte loads) and walks IT voxels whose loads are all issued before
// the first use; 32-bit element offsets from the batch item's base, no division per element.
struct Nb8Args {
  const float* dout; const float* y; float* o;
  long long dsn, ysn, osn;
  unsigned dsw, ysw, osw;
  int C, relu;
  unsigned dhw;
  const float* mean; const float* rstd; const float* gamma; const float* beta; const float* m1; const float* m2;
};

template <bool YBF, int IT, bool DBF = false>
__global__ __launch_bounds__(256) void norm_bwd_apply8_kernel(Nb8Args a) {
  const int n = blockIdx.y;
  const unsigned CG = (unsigned)a.C >> 3, nvl = 256u / CG;
  const unsigned cg = threadIdx.x & (CG - 1), vl = threadIdx.x / CG;   // CG is a power of two
  const unsigned c0 = cg * 8;
  const float* dp = item_base<DBF>(a.dout, n, a.dsn);                 // gradient in / out share one storage type (DBF)
  const float* yp = item_base<YBF>(a.y, n, a.ysn);
  float* op = const_cast<float*>(item_base<DBF>(a.o, n, a.osn));
  const unsigned v0 = blockIdx.x * (nvl * IT) + vl;
  Oct8<YBF> yr[IT];
  Oct8<DBF> dr[IT];
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const unsigned v = min(v0 + i * nvl, a.dhw - 1);                   // clamped address, masked store
    yr[i] = oct8_ld<YBF>(yp, v * a.ysw + c0, v * a.ysw + c0 + 4);
    dr[i] = oct8_ld<DBF>(dp, v * a.dsw + c0, v * a.dsw + c0 + 4);
  }
  float mu[8], rs[8], g[8], bt[8], m1[8], m2[8];
  {
    const unsigned pc = (unsigned)n * a.C + c0;
    auto ld8 = [](const float* p, unsigned off, float (&r)[8]) {
      const float4 lo = *reinterpret_cast<const float4*>(p + off), hi = *reinterpret_cast<const float4*>(p + off + 4);
      r[0] = lo.x; r[1] = lo.y; r[2] = lo.z; r[3] = lo.w; r[4] = hi.x; r[5] = hi.y; r[6] = hi.z; r[7] = hi.w;
    };
    ld8(a.mean, pc, mu); ld8(a.rstd, pc, rs); ld8(a.m1, pc, m1); ld8(a.m2, pc, m2);
    if (a.gamma) ld8(a.gamma, c0, g);
    else {
#pragma unroll
      for (int j = 0; j < 8; ++j) g[j] = 1.f;
    }
    if (a.beta) ld8(a.beta, c0, bt);
    else {
#pragma unroll
      for (int j = 0; j < 8; ++j) bt[j] = 0.f;
    }
  }
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    float yv[8], ov[8], dv[8];
    oct8_f8(yr[i], yv);
    oct8_f8(dr[i], dv);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float xhat = (yv[j] - mu[j]) * rs[j];
      float dz = dv[j];
      if (a.relu && !(fmaf(g[j], xhat, bt[j]) > 0.f)) dz = 0.f;
      ov[j] = rs[j] * (g[j] * dz - m1[j] - xhat * m2[j]);
    }
    const unsigned v = v0 + i * nvl;
    if (v < a.dhw) oct8_st<DBF>(op, v * a.osw + c0, ov);
  }
}

// ------------------------------------------------------------------ whole instance-norm backward of a small tensor
// The deep levels (<= 16^3 voxels) ran reduce -> finalize -> apply as three launches of a few microseconds each, all of
// them launch latency.  Here one workgroup owns 32 channels of one batch item: pass 1 sums dz and dz * xhat over all
// voxels (a thread keeps one channel octet, 64 voxel lanes), the 64 partials of a channel are added in a fixed order
// in fp64 (like instance_bwd_kernel), pass 2 re-reads the two tensors (L2 hits) and writes dy.  No affine gradients.
struct NbsArgs {
  const float* dout; const float* y; float* o;
  long long dsn, ysn, osn;
  unsigned dsw, ysw, osw;
  int C, relu;
  unsigned dhw;
  double count;
  const float* mean; const float* rstd; const float* gamma; const float* beta;
};

template <bool YBF, bool DBF = false>
__global__ __launch_bounds__(256) void norm_bwd_small_kernel(NbsArgs a) {
  __shared__ float part[2][8][256];
  __shared__ float mm[2][32];
  const int n = blockIdx.y;
  const unsigned cg = threadIdx.x & 3, vl = threadIdx.x >> 2;
  const unsigned c0 = blockIdx.x * 32 + cg * 8;
  const float* dp = item_base<DBF>(a.dout, n, a.dsn);
  const float* yp = item_base<YBF>(a.y, n, a.ysn);
  float* op = const_cast<float*>(item_base<DBF>(a.o, n, a.osn));
  float mu[8], rs[8], g[8], bt[8];
  {
    const unsigned pc = (unsigned)n * a.C + c0;
    auto ld8 = [](const float* p, unsigned off, float (&r)[8]) {
      const float4 lo = *reinterpret_cast<const float4*>(p + off), hi = *reinterpret_cast<const float4*>(p + off + 4);
      r[0] = lo.x; r[1] = lo.y; r[2] = lo.z; r[3] = lo.w; r[4] = hi.x; r[5] = hi.y; r[6] = hi.z; r[7] = hi.w;
    };
    ld8(a.mean, pc, mu); ld8(a.rstd, pc, rs);
    if (a.gamma) ld8(a.gamma, c0, g);
    else {
#pragma unroll
      for (int j = 0; j < 8; ++j) g[j] = 1.f;
    }
    if (a.beta) ld8(a.beta, c0, bt);
    else {
#pragma unroll
      for (int j = 0; j < 8; ++j) bt[j] = 0.f;
    }
  }
  constexpr int IT = 2;
  float s0[8], s1[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s0[j] = 0.f; s1[j] = 0.f; }
  for (unsigned vb = vl; vb < a.dhw; vb += 64 * IT) {
    Oct8<YBF> yr[IT];
    Oct8<DBF> dr[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const unsigned v = min(vb + 64 * i, a.dhw - 1);
      yr[i] = oct8_ld<YBF>(yp, v * a.ysw + c0, v * a.ysw + c0 + 4);
      dr[i] = oct8_ld<DBF>(dp, v * a.dsw + c0, v * a.dsw + c0 + 4);
    }
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      float yv[8], dv[8];
      oct8_f8(yr[i], yv);
      oct8_f8(dr[i], dv);
      const bool live = vb + 64 * i < a.dhw;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xhat = (yv[j] - mu[j]) * rs[j];
        float dz = live ? dv[j] : 0.f;
        if (a.relu && !(fmaf(g[j], xhat, bt[j]) > 0.f)) dz = 0.f;
        s0[j] += dz;
        s1[j] += dz * xhat;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) { part[0][j][threadIdx.x] = s0[j]; part[1][j][threadIdx.x] = s1[j]; }
  __syncthreads();
  if (threadIdx.x < 64) {                        // thread -> (k, octet, j): the 64 voxel-lane partials of one channel
    const int k = threadIdx.x >> 5, q = (threadIdx.x >> 3) & 3, j = threadIdx.x & 7;
    double t = 0.0;
    for (int l = 0; l < 64; ++l) t += (double)part[k][j][l * 4 + q];
    const int c = blockIdx.x * 32 + q * 8 + j;
    const double gm = a.gamma ? (double)a.gamma[c] : 1.0;
    mm[k][q * 8 + j] = (float)(gm * t / a.count);
  }
  __syncthreads();
  float m1[8], m2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { m1[j] = mm[0][cg * 8 + j]; m2[j] = mm[1][cg * 8 + j]; }
  for (unsigned vb = vl; vb < a.dhw; vb += 64 * IT) {
    Oct8<YBF> yr[IT];
    Oct8<DBF> dr[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const unsigned v = min(vb + 64 * i, a.dhw - 1);
      yr[i] = oct8_ld<YBF>(yp, v * a.ysw + c0, v * a.ysw + c0 + 4);
      dr[i] = oct8_ld<DBF>(dp, v * a.dsw + c0, v * a.dsw + c0 + 4);
    }
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      float yv[8], ov[8], dv[8];
      oct8_f8(yr[i], yv);
      oct8_f8(dr[i], dv);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xhat = (yv[j] - mu[j]) * rs[j];
        float dz = dv[j];
        if (a.relu && !(fmaf(g[j], xhat, bt[j]) > 0.f)) dz = 0.f;
        ov[j] = rs[j] * (g[j] * dz - m1[j] - xhat * m2[j]);
      }
      const unsigned v = vb + 64 * i;
      if (v < a.dhw) oct8_st<DBF>(op, v * a.osw + c0, ov);
    }
  }
}

// ------------------------------------------------------------------ combine, 8 channels per thread
// out = Ta(a) + Tb(b) for voxel-dense tensors of one storage type with C a power of two >= 8: the same arithmetic as
// elementwise_kernel<0, ...>, a thread's 16 coefficient pairs computed once, IT voxels' loads issued before the first use.
struct Cb8Args {
  const float* a; const float* b; float* o;
  long long asn, bsn, osn;
  unsigned asw, bsw, osw;
  int C;
  unsigned dhw;
  NL ta, tb;
};

template <bool BF, bool HASB, int IT>
__global__ __launch_bounds__(256) void combine8_kernel(Cb8Args e) {
  const int n = blockIdx.y;
  const unsigned CG = (unsigned)e.C >> 3, nvl = 256u / CG;
  const unsigned cg = threadIdx.x & (CG - 1), vl = threadIdx.x / CG;   // CG is a power of two
  const unsigned c0 = cg * 8;
  auto item = [&](const float* p, long long sn) {
    return BF ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(p) + (long long)n * sn) : p + (long long)n * sn;
  };
  const float* ap = item(e.a, e.asn);
  const float* bp = HASB ? item(e.b, e.bsn) : ap;
  float* op = const_cast<float*>(item(e.o, e.osn));
  const unsigned v0 = blockIdx.x * (nvl * IT) + vl;
  Oct8<BF> ar[IT], br[IT];
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const unsigned v = min(v0 + i * nvl, e.dhw - 1);                   // clamped address, masked store
    ar[i] = oct8_ld<BF>(ap, v * e.asw + c0, v * e.asw + c0 + 4);
    if (HASB) br[i] = oct8_ld<BF>(bp, v * e.bsw + c0, v * e.bsw + c0 + 4);
  }
  float sa[8], ha[8], sb[8], hb[8];
  nl_coeff_vec<8>(e.ta, n, e.C, (int)c0, sa, ha);
  if (HASB) nl_coeff_vec<8>(e.tb, n, e.C, (int)c0, sb, hb);
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    float av[8], bv[8], ov[8];
    oct8_f8(ar[i], av);
    if (HASB) oct8_f8(br[i], bv);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float r = nl_apply(av[j], sa[j], ha[j], e.ta.relu);
      if (HASB) r += nl_apply(bv[j], sb[j], hb[j], e.tb.relu);
      ov[j] = r;
    }
    const unsigned v = v0 + i * nvl;
    if (v < e.dhw) {
      if constexpr (BF) {
        uint4 q;
        q.x = f32x2_to_bf16x2(ov[0], ov[1]); q.y = f32x2_to_bf16x2(ov[2], ov[3]);
        q.z = f32x2_to_bf16x2(ov[4], ov[5]); q.w = f32x2_to_bf16x2(ov[6], ov[7]);
        *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(op) + v * e.osw + c0) = q;
      } else {
        *reinterpret_cast<float4*>(op + v * e.osw + c0) = make_float4(ov[0], ov[1], ov[2], ov[3]);
        *reinterpret_cast<float4*>(op + v * e.osw + c0 + 4) = make_float4(ov[4], ov[5], ov[6], ov[7]);
      }
    }
  }
}

template <bool BF, bool HASB>
static void launch_combine8(const Cb8Args& q, int n, hipStream_t s) {
  const long long nvl = 256 / (q.C / 8);
  const bool four = q.dhw / (nvl * 4) >= 1024;
  const long long per = nvl * (four ? 4 : 2);
  const dim3 g8((unsigned)((q.dhw + per - 1) / per), (unsigned)n);
  if (four) hipLaunchKernelGGL((combine8_kernel<BF, HASB, 4>), g8, dim3(256), 0, s, q);
  else hipLaunchKernelGGL((combine8_kernel<BF, HASB, 2>), g8, dim3(256), 0, s, q);
}

// ------------------------------------------------------------------ linear combination
struct LinArgs {
  TV in[8];
  float w[8];
  int count;
  TV o;
  int accumulate;
};

// COUNT inputs are a template parameter: their loads are straight-line (a load behind a run-time `k < count`
// branch is waited for before the next one is issued)
// IBF / OBF: storage of the inputs (all alike) and of the output (bf16: the wide forward activations of bf16 precision)
template <int VEC, int COUNT, bool IBF = false, bool OBF = false>
__global__ __launch_bounds__(256) void lincomb_kernel(LinArgs e) {
  const int C = e.o.c;
  const int CV = (C + VEC - 1) / VEC;
  const long long nvox = (long long)e.o.n * e.o.d * e.o.h * e.o.w;
  const long long total = nvox * CV;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c0 = (int)(i % CV) * VEC;
    const long long v = i / CV;
    int n, z, y, x;
    vox_decompose(e.o, v, n, z, y, x);
    float acc[VEC];
    const long long oo = vox_addr(e.o, n, z, y, x) + c0;          // element offsets: the same code addresses 2- and 4-byte elements
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
    if (e.accumulate) {
      if (VEC == 4) { const float4 t = ld4_t<OBF>(e.o.p, oo); acc[0] = t.x; acc[1] = t.y; acc[2] = t.z; acc[3] = t.w; }
      else acc[0] = ld1_t<OBF>(e.o.p, oo);
    }
    float4 tv4[COUNT];
    float ts[COUNT];
#pragma unroll
    for (int k = 0; k < COUNT; ++k) {
      const long long io = vox_addr(e.in[k], n, z, y, x) + c0;
      if (VEC == 4) tv4[k] = ld4_t<IBF>(e.in[k].p, io);
      else ts[k] = ld1_t<IBF>(e.in[k].p, io);
    }
#pragma unroll
    for (int k = 0; k < COUNT; ++k) {
      if (VEC == 4) {
        acc[0] = fmaf(e.w[k], tv4[k].x, acc[0]); acc[1] = fmaf(e.w[k], tv4[k].y, acc[1]);
        acc[2] = fmaf(e.w[k], tv4[k].z, acc[2]); acc[3] = fmaf(e.w[k], tv4[k].w, acc[3]);
      } else {
        acc[0] = fmaf(e.w[k], ts[k], acc[0]);
      }
    }
    if (VEC == 4) st4_t<OBF>(e.o.p, oo, make_float4(acc[0], acc[1], acc[2], acc[3]));
    else st1_t<OBF>(e.o.p, oo, acc[0]);
  }
}

// ------------------------------------------------------------------ trilinear x2 (align_corners=True)
__device__ __forceinline__ void src_index(int dst, int in, int out, int& i0, int& i1, float& lam) {
  // torch: scale = (in-1)/(out-1) (0 when out == 1); src = scale*dst; i0 = floor; i1 = min(i0+1, in-1)
  const float scale = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
  const float src = scale * (float)dst;
  i0 = (int)src;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  lam = src - (float)i0;
}

// VEC = 4: a thread owns 4 consecutive channels of one output voxel (16-byte accesses, index arithmetic amortised);
// VEC = 1: scalar fallback for unaligned views.
template <int VEC, bool BF = false>
__global__ __launch_bounds__(256) void upsample_fwd_kernel(TV x, TV y) {
  const int C = y.c;
  const int CV = (C + VEC - 1) / VEC;
  const long long total = (long long)y.n * y.d * y.h * y.w * CV;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % CV) * VEC;
    int n, oz, oy, ox;
    vox_decompose(y, i / CV, n, oz, oy, ox);
    int z0, z1, y0, y1, x0, x1;
    float lz, ly, lx;
    src_index(oz, x.d, y.d, z0, z1, lz);
    src_index(oy, x.h, y.h, y0, y1, ly);
    src_index(ox, x.w, y.w, x0, x1, lx);
    const float* xp = x.p;
    const long long xb = (long long)n * x.sn + c;                 // element offset of (item, channel): x and y share one storage (BF)
    const float wz0 = 1.f - lz, wy0 = 1.f - ly, wx0 = 1.f - lx;
    const long long o00 = (long long)z0 * x.sd + (long long)y0 * x.sh, o01 = (long long)z0 * x.sd + (long long)y1 * x.sh;
    const long long o10 = (long long)z1 * x.sd + (long long)y0 * x.sh, o11 = (long long)z1 * x.sd + (long long)y1 * x.sh;
    const long long a0 = (long long)x0 * x.sw, a1 = (long long)x1 * x.sw;
    const long long oo = vox_addr(y, n, oz, oy, ox) + c;
    if (VEC == 4) {
      const float4 v000 = ld4_t<BF>(xp, xb + o00 + a0), v001 = ld4_t<BF>(xp, xb + o00 + a1);
      const float4 v010 = ld4_t<BF>(xp, xb + o01 + a0), v011 = ld4_t<BF>(xp, xb + o01 + a1);
      const float4 v100 = ld4_t<BF>(xp, xb + o10 + a0), v101 = ld4_t<BF>(xp, xb + o10 + a1);
      const float4 v110 = ld4_t<BF>(xp, xb + o11 + a0), v111 = ld4_t<BF>(xp, xb + o11 + a1);
#define MMTTA_TRI(f) (wz0 * (wy0 * (wx0 * v000.f + lx * v001.f) + ly * (wx0 * v010.f + lx * v011.f)) + \
                      lz * (wy0 * (wx0 * v100.f + lx * v101.f) + ly * (wx0 * v110.f + lx * v111.f)))
      st4_t<BF>(y.p, oo, make_float4(MMTTA_TRI(x), MMTTA_TRI(y), MMTTA_TRI(z), MMTTA_TRI(w)));
#undef MMTTA_TRI
    } else {
      auto X = [&](long long o) { return ld1_t<BF>(xp, xb + o); };
      const float v = wz0 * (wy0 * (wx0 * X(o00 + a0) + lx * X(o00 + a1)) + ly * (wx0 * X(o01 + a0) + lx * X(o01 + a1))) +
                      lz * (wy0 * (wx0 * X(o10 + a0) + lx * X(o10 + a1)) + ly * (wx0 * X(o11 + a0) + lx * X(o11 + a1)));
      st1_t<BF>(y.p, oo, v);
    }
  }
}

// adjoint as a gather: for input index i the outputs that touch it lie in [2i-2, 2i+2]
__device__ __forceinline__ int axis_weights(int i, int in, int out, int* os, float* ws) {
  int cnt = 0;
  int lo = 2 * i - 2, hi = 2 * i + 2;
  if (lo < 0) lo = 0;
  if (hi > out - 1) hi = out - 1;
  for (int o = lo; o <= hi; ++o) {
    int i0, i1;
    float lam;
    src_index(o, in, out, i0, i1, lam);
    float w = 0.f;
    if (i0 == i) w += 1.f - lam;
    if (i1 == i) w += lam;   // i0 == i1 == i at the last sample: (1-lam) + lam = 1
    if (i0 == i1 && i0 == i) w = 1.f;
    if (w != 0.f) { os[cnt] = o; ws[cnt] = w; ++cnt; }
  }
  return cnt;
}

// BF: both gradients bf16-stored (method.grad_storage; offsets are elements either way)
template <int VEC, bool BF = false>
__global__ __launch_bounds__(256) void upsample_bwd_kernel(TV dy, TV dx, int accumulate) {
  const int C = dx.c;
  const int CV = (C + VEC - 1) / VEC;
  const long long total = (long long)dx.n * dx.d * dx.h * dx.w * CV;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % CV) * VEC;
    int n, iz, iy, ix;
    vox_decompose(dx, i / CV, n, iz, iy, ix);
    int oz[5], oy[5], ox[5];
    float wz[5], wy[5], wx[5];
    const int nz = axis_weights(iz, dx.d, dy.d, oz, wz);
    const int ny = axis_weights(iy, dx.h, dy.h, oy, wy);
    const int nx = axis_weights(ix, dx.w, dy.w, ox, wx);
    const long long gbase = (long long)n * dy.sn + c;
    float s[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) s[j] = 0.f;
    for (int a = 0; a < nz; ++a)
      for (int b = 0; b < ny; ++b) {
        const float wzy = wz[a] * wy[b];
        const long long row = gbase + (long long)oz[a] * dy.sd + (long long)oy[b] * dy.sh;
        for (int k = 0; k < nx; ++k) {
          const float wgt = wzy * wx[k];
          if (VEC == 4) {
            const float4 g4 = ld4_any(dy.p, row + (long long)ox[k] * dy.sw, BF);
            s[0] = fmaf(wgt, g4.x, s[0]); s[1] = fmaf(wgt, g4.y, s[1]); s[2] = fmaf(wgt, g4.z, s[2]); s[3] = fmaf(wgt, g4.w, s[3]);
          } else {
            s[0] += wgt * ld1_any(dy.p, row + (long long)ox[k] * dy.sw, BF);
          }
        }
      }
    const long long o = vox_addr(dx, n, iz, iy, ix) + c;
    if (VEC == 4) {
      float4 r = make_float4(s[0], s[1], s[2], s[3]);
      if (accumulate) { const float4 t = ld4_any(dx.p, o, BF); r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w; }
      st4_any(dx.p, o, r, BF);
    } else {
      st1_any(dx.p, o, accumulate ? ld1_any(dx.p, o, BF) + s[0] : s[0], BF);
    }
  }
}

static inline int grid_for(long long total, int cap = 8192) {
  long long b = (total + 255) / 256;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (int)b;
}

// 16-byte access to 4 channels at a time.  C need not be a multiple of 4 when the voxel row is padded to one
// (sw >= roundup(C,4)): READING the pad lanes is always harmless; WRITING them needs a view that owns its pad.
static inline bool vec4_rd(const mmtta_tensor* t) {
  return ((uintptr_t)t->ptr) % (t->dtype == MMTTA_BF16 ? 8 : 16) == 0 && t->sw % 4 == 0 && t->sh % 4 == 0 && t->sd % 4 == 0 &&
         t->sn % 4 == 0 && t->sc == 1 &&
         (t->c % 4 == 0 || t->sw >= (t->c + 3) / 4 * 4);
}
static inline bool vec4_wr(const mmtta_tensor* t) {
  return vec4_rd(t) && (t->c % 4 == 0 || (t->flags & MMTTA_TENSOR_OWNS_PAD));
}

static inline bool same_shape(const mmtta_tensor* a, const mmtta_tensor* b) {
  return a->n == b->n && a->c == b->c && a->d == b->d && a->h == b->h && a->w == b->w;
}

static void rows_geometry(const mmtta_tensor* t, int& rows_per_n, long long& vox_per_row) {
  const long long dhw = (long long)t->d * t->h * t->w;
  long long vpr = (dhw + 1023) / 1024;     // at most 1024 rows per batch item
  if (vpr < 32) vpr = 32;
  rows_per_n = (int)((dhw + vpr - 1) / vpr);
  if (rows_per_n < 1) rows_per_n = 1;
  vox_per_row = vpr;
}

int channel_partial_rows(const mmtta_tensor* t) {
  int r; long long v;
  rows_geometry(t, r, v);
  return r;
}

int launch_channel_sums(const mmtta_tensor* x, float* part, hipStream_t s) {
  RedArgs a;
  a.x = tv(x); a.dout = tv(x); a.t = nl(nullptr); a.part = part;
  rows_geometry(x, a.rows_per_n, a.vox_per_row);
  const dim3 grid(x->n * a.rows_per_n);
  if (is_bf16(x)) {
    if (vec4_rd(x)) hipLaunchKernelGGL((channel_reduce_kernel<0, 4, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((channel_reduce_kernel<0, 1, true>), grid, dim3(256), 0, s, a);
  } else {
    if (vec4_rd(x)) hipLaunchKernelGGL((channel_reduce_kernel<0, 4>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((channel_reduce_kernel<0, 1>), grid, dim3(256), 0, s, a);
  }
  return launch_status("channel sums");
}

}  // namespace mmtta

using namespace mmtta;

extern "C" int mmtta_copy_strided(const mmtta_tensor* src, const mmtta_tensor* dst, void* stream) {
  MMTTA_CHECK(src == nullptr || src->dtype == MMTTA_F32, MMTTA_ERR_UNSUPPORTED, "mmtta_copy_strided: `src` must be fp32-stored");
  MMTTA_CHECK(src && dst && src->ptr && dst->ptr, MMTTA_ERR_INVALID, "copy: null tensor");
  MMTTA_CHECK(same_shape(src, dst), MMTTA_ERR_INVALID, "copy: shape mismatch");
  const long long total = (long long)dst->n * dst->c * dst->d * dst->h * dst->w;
  // (dst may be bf16-stored: the network input of bf16 precision is rounded once here instead of by every consumer)
  if (is_bf16(dst)) hipLaunchKernelGGL(copy_strided_kernel<true>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, tv(src), tv(dst));
  else hipLaunchKernelGGL(copy_strided_kernel<false>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, tv(src), tv(dst));
  return launch_status("copy_strided");
}

extern "C" int mmtta_reduce_rows_per_n(const mmtta_tensor* t) {
  if (t == nullptr) return -1;
  int r; long long v;
  rows_geometry(t, r, v);
  return r;
}

extern "C" int mmtta_channel_stats(const mmtta_tensor* x, float* part, void* stream) {
  MMTTA_CHECK(x && x->ptr && part, MMTTA_ERR_INVALID, "channel_stats: null argument");
  MMTTA_CHECK(is_cl(x), MMTTA_ERR_UNSUPPORTED, "channel_stats: tensor must be channels-last");
  return launch_channel_sums(x, part, (hipStream_t)stream);
}

extern "C" int mmtta_norm_stats_finalize(int kind, int groups, const float* part, int rows_per_n, int n, int c,
                                         int64_t count, float eps, int training, float* running_mean,
                                         float* running_var, float momentum, float* mean, float* rstd,
                                         const float* gamma, const float* beta, float* scale, float* shift,
                                         double* scratch, void* stream) {
  MMTTA_CHECK(kind >= 0 && kind <= 2, MMTTA_ERR_INVALID, "norm: bad kind %d", kind);
  MMTTA_CHECK(mean && rstd && n > 0 && c > 0 && count > 0, MMTTA_ERR_INVALID, "norm finalize: bad argument");
  MMTTA_CHECK((scale == nullptr) == (shift == nullptr), MMTTA_ERR_INVALID, "norm finalize: scale and shift go together");
  MMTTA_CHECK(scratch != nullptr, MMTTA_ERR_INVALID, "norm finalize: null scratch");
  double* g_tot = scratch;
  if (kind == MMTTA_NORM_GROUP) MMTTA_CHECK(groups > 0 && c % groups == 0, MMTTA_ERR_INVALID, "group norm: C %% groups != 0");
  if (kind == MMTTA_NORM_BATCH && !training) MMTTA_CHECK(running_mean && running_var, MMTTA_ERR_INVALID, "batch norm eval needs running stats");
  hipStream_t s = (hipStream_t)stream;
  if (kind == MMTTA_NORM_INSTANCE) {
    MMTTA_CHECK(part != nullptr && rows_per_n > 0, MMTTA_ERR_INVALID, "norm finalize: null partials");
    hipLaunchKernelGGL(instance_stats_kernel, dim3(n * c), dim3(64), 0, s, part, rows_per_n, c, (double)count, eps, mean, rstd,
                       gamma, beta, scale, shift);
    return launch_status("instance norm stats");
  }
  const bool need_rows = !(kind == MMTTA_NORM_BATCH && !training);
  if (need_rows) {
    MMTTA_CHECK(part != nullptr && rows_per_n > 0, MMTTA_ERR_INVALID, "norm finalize: null partials");
    hipLaunchKernelGGL(rows_reduce_kernel, dim3(n * c), dim3(64), 0, s, part, rows_per_n, c, g_tot);
    int st = launch_status("norm rows reduce");
    if (st) return st;
  }
  StatFinArgs a;
  a.kind = kind; a.groups = groups; a.N = n; a.C = c; a.count = (double)count; a.eps = eps; a.training = training;
  a.running_mean = running_mean; a.running_var = running_var; a.momentum = momentum; a.mean = mean; a.rstd = rstd;
  a.gamma = gamma; a.beta = beta; a.scale = scale; a.shift = shift;
  a.tot = g_tot;
  hipLaunchKernelGGL(stats_finalize_kernel, dim3((n * c + 63) / 64), dim3(64), 0, s, a);
  return launch_status("norm stats finalize");
}

extern "C" int mmtta_combine(const mmtta_tensor* a, const mmtta_norm_on_load* ta, const mmtta_tensor* b,
                             const mmtta_norm_on_load* tb, const mmtta_tensor* out, void* stream) {
  MMTTA_CHECK(a && out && a->ptr && out->ptr, MMTTA_ERR_INVALID, "combine: null tensor");
  MMTTA_CHECK(same_shape(a, out) && (!b || same_shape(b, out)), MMTTA_ERR_INVALID, "combine: shape mismatch");
  MMTTA_CHECK(is_cl(a) && is_cl(out) && (!b || is_cl(b)), MMTTA_ERR_UNSUPPORTED, "combine: channels-last only");
  EwArgs e;
  e.a = tv(a); e.b = b ? tv(b) : tv(a); e.o = tv(out); e.ta = nl(ta); e.tb = nl(tb); e.m1 = e.m2 = nullptr; e.hasb = b ? 1 : 0;
  const bool v4 = vec4_rd(a) && vec4_wr(out) && (!b || vec4_rd(b));
  const long long total = (long long)out->n * out->d * out->h * out->w * (v4 ? (out->c + 3) / 4 : out->c);
  // storage: all fp32, or all bf16 (the wide forward activations of bf16 precision)
  const bool abf = is_bf16(a), bbf = b ? is_bf16(b) : abf, obf = is_bf16(out);
  MMTTA_CHECK((abf == bbf && bbf == obf), MMTTA_ERR_UNSUPPORTED, "combine: operands must share one storage type");
  const dim3 grid(grid_for(total));
  hipStream_t s = (hipStream_t)stream;
  {
    // octet form: C a power of two in [8, 2048], voxel-dense tensors, 16-byte aligned octets, 32-bit offsets inside an item
    const int C = out->c;
    const long long dhw = (long long)out->d * out->h * out->w;
    auto dense = [](const mmtta_tensor* t) { return t->sh == (int64_t)t->w * t->sw && t->sd == (int64_t)t->h * t->sh; };
    const int per = abf ? 8 : 4;
    auto al = [per](const mmtta_tensor* t) { return ((uintptr_t)t->ptr) % 16 == 0 && t->sw % per == 0 && t->sn % per == 0; };
    const bool pow2 = C >= 8 && C <= 2048 && (C & (C - 1)) == 0;
    const bool ok8 = v4 && pow2 && dense(a) && dense(out) && al(a) && al(out) && (!b || (dense(b) && al(b))) &&
                     dhw * std::max(std::max(a->sw, out->sw), b ? b->sw : (int64_t)0) < (1LL << 31);
    if (ok8) {
      Cb8Args q;
      q.a = (const float*)a->ptr; q.b = b ? (const float*)b->ptr : nullptr; q.o = (float*)out->ptr;
      q.asn = a->sn; q.bsn = b ? b->sn : 0; q.osn = out->sn;
      q.asw = (unsigned)a->sw; q.bsw = b ? (unsigned)b->sw : 0u; q.osw = (unsigned)out->sw;
      q.C = C; q.dhw = (unsigned)dhw; q.ta = e.ta; q.tb = e.tb;
      if (abf) { if (b) launch_combine8<true, true>(q, out->n, s); else launch_combine8<true, false>(q, out->n, s); }
      else { if (b) launch_combine8<false, true>(q, out->n, s); else launch_combine8<false, false>(q, out->n, s); }
      return launch_status("combine");
    }
  }
  if (abf) {
    if (v4) hipLaunchKernelGGL((elementwise_kernel<0, 4, true, true, true>), grid, dim3(256), 0, s, e);
    else hipLaunchKernelGGL((elementwise_kernel<0, 1, true, true, true>), grid, dim3(256), 0, s, e);
  } else {
    if (v4) hipLaunchKernelGGL((elementwise_kernel<0, 4>), grid, dim3(256), 0, s, e);
    else hipLaunchKernelGGL((elementwise_kernel<0, 1>), grid, dim3(256), 0, s, e);
  }
  return launch_status("combine");
}

extern "C" int mmtta_norm_bwd_reduce(const mmtta_tensor* dout, const mmtta_tensor* y, const mmtta_norm_on_load* t,
                                     float* part, void* stream) {
  MMTTA_CHECK(dout && y && t && part && dout->ptr && y->ptr && t->mean && t->rstd, MMTTA_ERR_INVALID, "norm bwd reduce: null argument");
  // (a bf16-stored gradient sits next to a bf16-stored activation - or, <= 4 channels, next to an fp32-stored one: the thin
  // full-resolution tensors keep their activations fp32, round 3)
  MMTTA_CHECK(!is_bf16(dout) || is_bf16(y) || y->c <= 4, MMTTA_ERR_UNSUPPORTED, "norm bwd reduce: a bf16-stored gradient needs a bf16-stored activation");
  MMTTA_CHECK(same_shape(dout, y) && is_cl(dout) && is_cl(y), MMTTA_ERR_INVALID, "norm bwd reduce: shape/layout mismatch");
  RedArgs a;
  a.x = tv(y); a.dout = tv(dout); a.t = nl(t); a.part = part;
  rows_geometry(y, a.rows_per_n, a.vox_per_row);
  const dim3 grid(y->n * a.rows_per_n);
  hipStream_t s = (hipStream_t)stream;
  const bool v4 = vec4_rd(y) && vec4_rd(dout);
  if (is_bf16(dout) && !is_bf16(y)) {
    if (v4) hipLaunchKernelGGL((channel_reduce_kernel<1, 4, false, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((channel_reduce_kernel<1, 1, false, true>), grid, dim3(256), 0, s, a);
  } else if (is_bf16(dout)) {
    if (v4) hipLaunchKernelGGL((channel_reduce_kernel<1, 4, true, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((channel_reduce_kernel<1, 1, true, true>), grid, dim3(256), 0, s, a);
  } else if (is_bf16(y)) {
    if (v4) hipLaunchKernelGGL((channel_reduce_kernel<1, 4, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((channel_reduce_kernel<1, 1, true>), grid, dim3(256), 0, s, a);
  } else {
    if (v4) hipLaunchKernelGGL((channel_reduce_kernel<1, 4>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((channel_reduce_kernel<1, 1>), grid, dim3(256), 0, s, a);
  }
  return launch_status("norm bwd reduce");
}

extern "C" int mmtta_norm_bwd_finalize(int kind, int groups, const float* part, int rows_per_n, int n, int c,
                                       int64_t count, const float* gamma, int training, float* m1, float* m2,
                                       float* dgamma, float* dbeta, int accumulate, double* scratch, void* stream) {
  MMTTA_CHECK(kind >= 0 && kind <= 2, MMTTA_ERR_INVALID, "norm: bad kind %d", kind);
  MMTTA_CHECK(part && m1 && m2 && n > 0 && c > 0 && count > 0 && rows_per_n > 0, MMTTA_ERR_INVALID, "norm bwd finalize: bad argument");
  MMTTA_CHECK(scratch != nullptr, MMTTA_ERR_INVALID, "norm bwd finalize: null scratch");
  double* g_tot = scratch;
  if (kind == MMTTA_NORM_GROUP) MMTTA_CHECK(groups > 0 && c % groups == 0, MMTTA_ERR_INVALID, "group norm: C %% groups != 0");
  hipStream_t s = (hipStream_t)stream;
  if (kind == MMTTA_NORM_INSTANCE && dgamma == nullptr) {
    hipLaunchKernelGGL(instance_bwd_kernel, dim3(n * c), dim3(64), 0, s, part, rows_per_n, c, (double)count, gamma, m1, m2);
    return launch_status("instance norm bwd finalize");
  }
  hipLaunchKernelGGL(rows_reduce_kernel, dim3(n * c), dim3(64), 0, s, part, rows_per_n, c, g_tot);
  int st = launch_status("norm bwd rows reduce");
  if (st) return st;
  BwdFinArgs a;
  a.kind = kind; a.groups = groups; a.N = n; a.C = c; a.count = (double)count; a.gamma = gamma; a.training = training;
  a.m1 = m1; a.m2 = m2; a.dgamma = dgamma; a.dbeta = dbeta; a.accumulate = accumulate; a.tot = g_tot;
  hipLaunchKernelGGL(bwd_finalize_kernel, dim3((n * c + 63) / 64), dim3(64), 0, s, a);
  return launch_status("norm bwd finalize");
}

extern "C" int mmtta_norm_bwd_apply(const mmtta_tensor* dout, const mmtta_tensor* y, const mmtta_norm_on_load* t,
                                    const float* m1, const float* m2, const mmtta_tensor* dy, void* stream) {
  MMTTA_CHECK(dout && y && t && dy && m1 && m2 && t->mean && t->rstd, MMTTA_ERR_INVALID, "norm bwd apply: null argument");
  MMTTA_CHECK(dout->dtype == dy->dtype && (!is_bf16(dout) || is_bf16(y) || y->c <= 4), MMTTA_ERR_UNSUPPORTED,
              "norm bwd apply: `dout` and `dy` share one storage type (bf16 only next to a bf16-stored activation)");
  const bool dbf = is_bf16(dout);
  MMTTA_CHECK(same_shape(dout, y) && same_shape(dy, y), MMTTA_ERR_INVALID, "norm bwd apply: shape mismatch");
  MMTTA_CHECK(is_cl(dout) && is_cl(y) && is_cl(dy), MMTTA_ERR_UNSUPPORTED, "norm bwd apply: channels-last only");
  EwArgs e;
  e.a = tv(dout); e.b = tv(y); e.o = tv(dy); e.ta = nl(t); e.tb = nl(nullptr); e.m1 = m1; e.m2 = m2; e.hasb = 1;
  const bool v4 = vec4_rd(dout) && vec4_rd(y) && vec4_wr(dy);
  const long long total = (long long)y->n * y->d * y->h * y->w * (v4 ? (y->c + 3) / 4 : y->c);
  const dim3 grid(grid_for(total));
  hipStream_t s = (hipStream_t)stream;
  {
    // octet form: C a power of two in [8, 2048], voxel-dense tensors, 16-byte aligned octets, 32-bit offsets inside an item
    const int C = y->c;
    const long long dhw = (long long)y->d * y->h * y->w;
    auto dense = [](const mmtta_tensor* t) { return t->sh == (int64_t)t->w * t->sw && t->sd == (int64_t)t->h * t->sh; };
    auto al = [](const mmtta_tensor* t, int per) {
      return ((uintptr_t)t->ptr) % 16 == 0 && t->sw % per == 0 && t->sn % per == 0;
    };
    const bool pow2 = C >= 8 && C <= 2048 && (C & (C - 1)) == 0;
    const bool ok8 = v4 && pow2 && dense(dout) && dense(y) && dense(dy) && al(dout, dbf ? 8 : 4) && al(dy, dbf ? 8 : 4) &&
                     al(y, is_bf16(y) ? 8 : 4) &&
                     ((uintptr_t)t->mean % 16 == 0) && ((uintptr_t)t->rstd % 16 == 0) && ((uintptr_t)m1 % 16 == 0) &&
                     ((uintptr_t)m2 % 16 == 0) && (!t->gamma || (uintptr_t)t->gamma % 16 == 0) &&
                     (!t->beta || (uintptr_t)t->beta % 16 == 0) && dhw * std::max(std::max(dout->sw, y->sw), dy->sw) < (1LL << 31);
    if (ok8) {
      Nb8Args q;
      q.dout = (const float*)dout->ptr; q.y = (const float*)y->ptr; q.o = (float*)dy->ptr;
      q.dsn = dout->sn; q.ysn = y->sn; q.osn = dy->sn;
      q.dsw = (unsigned)dout->sw; q.ysw = (unsigned)y->sw; q.osw = (unsigned)dy->sw;
      q.C = C; q.relu = t->relu; q.dhw = (unsigned)dhw;
      q.mean = t->mean; q.rstd = t->rstd; q.gamma = t->gamma; q.beta = t->beta; q.m1 = m1; q.m2 = m2;
      const long long nvl = 256 / (C / 8);
      const bool four = dhw / (nvl * 4) >= 1024;
      const long long per = nvl * (four ? 4 : 2);
      const dim3 g8((unsigned)((dhw + per - 1) / per), (unsigned)y->n);
      if (dbf) {
        if (four) hipLaunchKernelGGL((norm_bwd_apply8_kernel<true, 4, true>), g8, dim3(256), 0, s, q);
        else hipLaunchKernelGGL((norm_bwd_apply8_kernel<true, 2, true>), g8, dim3(256), 0, s, q);
      } else if (is_bf16(y)) {
        if (four) hipLaunchKernelGGL((norm_bwd_apply8_kernel<true, 4>), g8, dim3(256), 0, s, q);
        else hipLaunchKernelGGL((norm_bwd_apply8_kernel<true, 2>), g8, dim3(256), 0, s, q);
      } else {
        if (four) hipLaunchKernelGGL((norm_bwd_apply8_kernel<false, 4>), g8, dim3(256), 0, s, q);
        else hipLaunchKernelGGL((norm_bwd_apply8_kernel<false, 2>), g8, dim3(256), 0, s, q);
      }
      return launch_status("norm bwd apply");
    }
  }
  if (dbf && !is_bf16(y)) {          // thin tensors: bf16-stored gradients next to an fp32-stored activation
    if (v4) hipLaunchKernelGGL((elementwise_kernel<1, 4, true, false, true>), grid, dim3(256), 0, s, e);
    else hipLaunchKernelGGL((elementwise_kernel<1, 1, true, false, true>), grid, dim3(256), 0, s, e);
  } else if (dbf) {
    if (v4) hipLaunchKernelGGL((elementwise_kernel<1, 4, true, true, true>), grid, dim3(256), 0, s, e);
    else hipLaunchKernelGGL((elementwise_kernel<1, 1, true, true, true>), grid, dim3(256), 0, s, e);
  } else if (is_bf16(y)) {
    if (v4) hipLaunchKernelGGL((elementwise_kernel<1, 4, false, true, false>), grid, dim3(256), 0, s, e);
    else hipLaunchKernelGGL((elementwise_kernel<1, 1, false, true, false>), grid, dim3(256), 0, s, e);
  } else {
    if (v4) hipLaunchKernelGGL((elementwise_kernel<1, 4>), grid, dim3(256), 0, s, e);
    else hipLaunchKernelGGL((elementwise_kernel<1, 1>), grid, dim3(256), 0, s, e);
  }
  return launch_status("norm bwd apply");
}

// eligibility of the one-launch backward (host-only): instance statistics of a small voxel-dense tensor, channels in whole
// groups of 32, 16-byte aligned octets
static bool nbs_ok(const mmtta_tensor* dout, const mmtta_tensor* y, const mmtta_norm_on_load* t, const mmtta_tensor* dy) {
  if (!dout || !y || !t || !dy || !t->mean || !t->rstd) return false;
  if (dout->dtype != dy->dtype || (is_bf16(dout) && !is_bf16(y))) return false;
  if (!same_shape(dout, y) || !same_shape(dy, y) || !is_cl(dout) || !is_cl(y) || !is_cl(dy)) return false;
  const long long dhw = (long long)y->d * y->h * y->w;
  auto dense = [](const mmtta_tensor* x) { return x->sh == (int64_t)x->w * x->sw && x->sd == (int64_t)x->h * x->sh; };
  auto al = [](const mmtta_tensor* x, int per) { return ((uintptr_t)x->ptr) % 16 == 0 && x->sw % per == 0 && x->sn % per == 0; };
  auto a16 = [](const void* p) { return p == nullptr || ((uintptr_t)p) % 16 == 0; };
  return y->c % 32 == 0 && dhw >= 1 && dhw <= 4096 && dense(dout) && dense(y) && dense(dy) && al(dout, is_bf16(dout) ? 8 : 4) &&
         al(dy, is_bf16(dy) ? 8 : 4) && al(y, is_bf16(y) ? 8 : 4) && a16(t->mean) && a16(t->rstd) && a16(t->gamma) && a16(t->beta);
}

extern "C" int mmtta_norm_bwd_small_ok(const mmtta_tensor* dout, const mmtta_tensor* y, const mmtta_norm_on_load* t,
                                       const mmtta_tensor* dy) {
  return nbs_ok(dout, y, t, dy) ? 1 : 0;
}

extern "C" int mmtta_norm_bwd_small(const mmtta_tensor* dout, const mmtta_tensor* y, const mmtta_norm_on_load* t,
                                    int64_t count, const mmtta_tensor* dy, void* stream) {
  MMTTA_CHECK(nbs_ok(dout, y, t, dy), MMTTA_ERR_UNSUPPORTED, "norm bwd (one launch): tensors not eligible (mmtta_norm_bwd_small_ok)");
  MMTTA_CHECK(count > 0, MMTTA_ERR_INVALID, "norm bwd (one launch): count must be positive");
  NbsArgs q;
  q.dout = (const float*)dout->ptr; q.y = (const float*)y->ptr; q.o = (float*)dy->ptr;
  q.dsn = dout->sn; q.ysn = y->sn; q.osn = dy->sn;
  q.dsw = (unsigned)dout->sw; q.ysw = (unsigned)y->sw; q.osw = (unsigned)dy->sw;
  q.C = y->c; q.relu = t->relu; q.dhw = (unsigned)((long long)y->d * y->h * y->w); q.count = (double)count;
  q.mean = t->mean; q.rstd = t->rstd; q.gamma = t->gamma; q.beta = t->beta;
  const dim3 grid((unsigned)(y->c / 32), (unsigned)y->n);
  if (is_bf16(dout)) hipLaunchKernelGGL((norm_bwd_small_kernel<true, true>), grid, dim3(256), 0, (hipStream_t)stream, q);
  else if (is_bf16(y)) hipLaunchKernelGGL(norm_bwd_small_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, q);
  else hipLaunchKernelGGL(norm_bwd_small_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, q);
  return launch_status("norm bwd (one launch)");
}

extern "C" int mmtta_upsample2x_fwd(const mmtta_tensor* x, const mmtta_tensor* y, void* stream) {
  MMTTA_CHECK(x && y && x->ptr && y->ptr, MMTTA_ERR_INVALID, "upsample: null tensor");
  MMTTA_CHECK(x->dtype == y->dtype, MMTTA_ERR_UNSUPPORTED, "mmtta_upsample2x_fwd: `x` and `y` must share one storage type");
  MMTTA_CHECK(y->n == x->n && y->c == x->c && y->d == 2 * x->d && y->h == 2 * x->h && y->w == 2 * x->w, MMTTA_ERR_INVALID,
              "upsample: y must be exactly 2x of x");
  MMTTA_CHECK(is_cl(x) && is_cl(y), MMTTA_ERR_UNSUPPORTED, "upsample: channels-last only");
  const bool v4 = vec4_rd(x) && vec4_wr(y);
  const long long total = (long long)y->n * y->d * y->h * y->w * (v4 ? (y->c + 3) / 4 : y->c);
  const dim3 ug(grid_for(total, 16384));
  if (is_bf16(x)) {
    if (v4) hipLaunchKernelGGL((upsample_fwd_kernel<4, true>), ug, dim3(256), 0, (hipStream_t)stream, tv(x), tv(y));
    else hipLaunchKernelGGL((upsample_fwd_kernel<1, true>), ug, dim3(256), 0, (hipStream_t)stream, tv(x), tv(y));
  } else {
    if (v4) hipLaunchKernelGGL(upsample_fwd_kernel<4>, ug, dim3(256), 0, (hipStream_t)stream, tv(x), tv(y));
    else hipLaunchKernelGGL(upsample_fwd_kernel<1>, ug, dim3(256), 0, (hipStream_t)stream, tv(x), tv(y));
  }
  return launch_status("upsample fwd");
}

extern "C" int mmtta_upsample2x_bwd(const mmtta_tensor* dy, const mmtta_tensor* dx, int accumulate, void* stream) {
  MMTTA_CHECK(dx && dy && dx->ptr && dy->ptr, MMTTA_ERR_INVALID, "upsample bwd: null tensor");
  MMTTA_CHECK(dy->dtype == dx->dtype, MMTTA_ERR_UNSUPPORTED, "mmtta_upsample2x_bwd: `dy` and `dx` must share one storage type");
  MMTTA_CHECK(dy->n == dx->n && dy->c == dx->c && dy->d == 2 * dx->d && dy->h == 2 * dx->h && dy->w == 2 * dx->w,
              MMTTA_ERR_INVALID, "upsample bwd: dy must be exactly 2x of dx");
  MMTTA_CHECK(is_cl(dx) && is_cl(dy), MMTTA_ERR_UNSUPPORTED, "upsample bwd: channels-last only");
  const bool v4 = vec4_rd(dy) && vec4_wr(dx);
  const long long total = (long long)dx->n * dx->d * dx->h * dx->w * (v4 ? (dx->c + 3) / 4 : dx->c);
  const dim3 ug(grid_for(total, 16384));
  if (is_bf16(dy)) {
    if (v4) hipLaunchKernelGGL((upsample_bwd_kernel<4, true>), ug, dim3(256), 0, (hipStream_t)stream, tv(dy), tv(dx), accumulate);
    else hipLaunchKernelGGL((upsample_bwd_kernel<1, true>), ug, dim3(256), 0, (hipStream_t)stream, tv(dy), tv(dx), accumulate);
  } else {
    if (v4) hipLaunchKernelGGL(upsample_bwd_kernel<4>, ug, dim3(256), 0, (hipStream_t)stream, tv(dy), tv(dx), accumulate);
    else hipLaunchKernelGGL(upsample_bwd_kernel<1>, ug, dim3(256), 0, (hipStream_t)stream, tv(dy), tv(dx), accumulate);
  }
  return launch_status("upsample bwd");
}

extern "C" int mmtta_lincomb(int count, const mmtta_tensor* const* in, const float* w, const mmtta_tensor* out,
                             int accumulate, void* stream) {
  MMTTA_CHECK(count >= 1 && count <= 8 && in && w && out && out->ptr, MMTTA_ERR_INVALID, "lincomb: bad argument");
  for (int i = 0; i < count; ++i)
    MMTTA_CHECK(in[i] != nullptr && in[i]->dtype == in[0]->dtype, MMTTA_ERR_UNSUPPORTED, "lincomb: the inputs must share one storage type");
  const bool ibf = is_bf16(in[0]), obf = is_bf16(out);
  LinArgs e;
  bool v4 = vec4_wr(out);
  for (int k = 0; k < count; ++k) {
    MMTTA_CHECK(in[k] && in[k]->ptr && same_shape(in[k], out) && is_cl(in[k]), MMTTA_ERR_INVALID, "lincomb: input %d mismatch", k);
    e.in[k] = tv(in[k]);
    e.w[k] = w[k];
    v4 = v4 && vec4_rd(in[k]);
  }
  for (int k = count; k < 8; ++k) { e.in[k] = e.in[0]; e.w[k] = 0.f; }
  MMTTA_CHECK(is_cl(out), MMTTA_ERR_UNSUPPORTED, "lincomb: channels-last only");
  e.count = count; e.o = tv(out); e.accumulate = accumulate;
  const long long total = (long long)out->n * out->d * out->h * out->w * (v4 ? (out->c + 3) / 4 : out->c);
  const dim3 grid(grid_for(total)), block(256);
  hipStream_t st = (hipStream_t)stream;
#define MMTTA_LINCOMB_S(N, I, O) \
  do { if (v4) hipLaunchKernelGGL((lincomb_kernel<4, N, I, O>), grid, block, 0, st, e); \
       else hipLaunchKernelGGL((lincomb_kernel<1, N, I, O>), grid, block, 0, st, e); } while (0)
#define MMTTA_LINCOMB(N) \
  case N: if (ibf && obf) MMTTA_LINCOMB_S(N, true, true); else if (obf) MMTTA_LINCOMB_S(N, false, true); \
          else if (ibf) MMTTA_LINCOMB_S(N, true, false); else MMTTA_LINCOMB_S(N, false, false); break;
  switch (count) {
    MMTTA_LINCOMB(1) MMTTA_LINCOMB(2) MMTTA_LINCOMB(3) MMTTA_LINCOMB(4)
    MMTTA_LINCOMB(5) MMTTA_LINCOMB(6) MMTTA_LINCOMB(7) MMTTA_LINCOMB(8)
  }
#undef MMTTA_LINCOMB
#undef MMTTA_LINCOMB_S
  return launch_status("lincomb");
}
