// Weight / bias gradient of Conv3d and ConvTranspose3d on the CDNA4 matrix cores (gfx950).
//
// Replaces autograd's weight-gradient kernels reached from reference
// src/core/trainers/seg_trainer.py:142 (loss.backward()) for every conv of
// src/models/unet.py / src/models/unet_multimodal_midfusion.py.
//
// Both modules reduce to one form.  With a DENSE tensor on the coarse grid (Conv3d: dy,
// ConvTranspose3d: x) and a GATHERED tensor on the fine grid (Conv3d: x, ConvTranspose3d: dy):
//
//     R[tap][cg][cd] = sum_{n,o} T(G[n][o*s + tap - 1][cg]) * Td(D[n][o][cd])
//
// and the torch weight layout is [cd][cg][tap] for both (Conv3d [Cout,Cin,k], ConvT [Cin,Cout,k]).
// The reduction index is the voxel, so this is a GEMM with K = voxels:  A[row=cg][k=voxel],
// B[k=voxel][col=cd] on v_mfma_f32_32x32x2_f32.  One workgroup: one (32 cg x 32 cd) block, a run
// of spatial tiles; G's halo box and D's tile are staged in LDS once per tile and shared by all
// 27 taps; the four waves split the taps (7/7/7/6), i.e. 7 accumulators x 16 registers per lane.
// Partial results of the spatial splits go to a slab and are summed by a deterministic reduce
// kernel (no float atomics) that also emits the torch layout.
#include "common.h"

namespace mmtta {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WArgs {
  const float* g; long long gsn, gsd, gsh, gsw; int Cg, Dgg, Hgg, Wgg; NL tg;
  const float* dn; long long dsn, dsd, dsh, dsw; int Cd, Dd, Hd, Wd; NL td;
  int N;
  int si, ntaps;
  float* slab;     // [nsl][ntaps][CGp][CDp]
  float* dbpart;   // [nsl][CDp] or null
  int tz, ty, tx, tiles, tiles_per_split;
  int CGp, CDp;
  int gvec4, dvec4;
};

template <int TZ, int TY, int TX>
__global__ __launch_bounds__(256) void wgrad_f32_kernel(WArgs a) {
  extern __shared__ float lds[];
  constexpr int MT = TZ * TY * TX;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, r = lane & 31;
  const int ext = a.ntaps == 1 ? 0 : 2;
  const int BZ = (TZ - 1) * a.si + ext + 1, BY = (TY - 1) * a.si + ext + 1, BX = (TX - 1) * a.si + ext + 1;
  const int boxvox = BZ * BY * BX;
  const int dmin = a.ntaps == 1 ? 0 : -1;
  float* gl = lds;                 // [boxvox][32]
  float* dl = lds + boxvox * 32;   // [MT][32]
  const int cg0 = blockIdx.y * 32, cd0 = blockIdx.z * 32;

  // taps of this wave: tap = wave + 4*j (27 taps) ; single tap: every wave, k-steps interleaved
  int toff[7];
  int ntw = 0;
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    const int tap = wave + 4 * j;
    toff[j] = 0;
    if (a.ntaps == 1) { if (j == 0) ntw = 1; }
    else if (tap < 27) { toff[j] = ((tap / 9) * BY + ((tap / 3) % 3)) * BX + (tap % 3); ntw = j + 1; }
  }
  const int kstep0 = a.ntaps == 1 ? wave : 0, kstride = a.ntaps == 1 ? 4 : 1;

  f32x16 acc[7];
#pragma unroll
  for (int j = 0; j < 7; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
  float dbacc = 0.f;

  const int t0 = blockIdx.x * a.tiles_per_split;
  const int t1 = min(a.tiles, t0 + a.tiles_per_split);
  const int tpn = a.tz * a.ty * a.tx;
  for (int tile = t0; tile < t1; ++tile) {
    const int n = tile / tpn;
    int t = tile % tpn;
    const int txi = t % a.tx; t /= a.tx;
    const int tyi = t % a.ty;
    const int tzi = t / a.ty;
    const int oz0 = tzi * TZ, oy0 = tyi * TY, ox0 = txi * TX;
    const int iz0 = oz0 * a.si + dmin, iy0 = oy0 * a.si + dmin, ix0 = ox0 * a.si + dmin;
    // ---- stage G box: [boxvox][32 channels cg0..cg0+31]
    {
      const float* gb = a.g + (long long)n * a.gsn;
      if (a.gvec4) {
        const int cv = tid & 7;
        const int c = cg0 + cv * 4;
        float sc[4], sh[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (c + j < a.Cg) nl_coeff(a.tg, n, a.Cg, c + j, sc[j], sh[j]);
          else { sc[j] = 0.f; sh[j] = 0.f; }
        }
        for (int bv = tid >> 3; bv < boxvox; bv += 32) {
          const int bx = bv % BX, by = (bv / BX) % BY, bz = bv / (BX * BY);
          const int iz = iz0 + bz, iy = iy0 + by, ix = ix0 + bx;
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if ((unsigned)iz < (unsigned)a.Dgg && (unsigned)iy < (unsigned)a.Hgg && (unsigned)ix < (unsigned)a.Wgg &&
              c < a.Cg) {
            const float4 x = *reinterpret_cast<const float4*>(gb + iz * a.gsd + iy * a.gsh + ix * a.gsw + c);
            v.x = nl_apply(x.x, sc[0], sh[0], a.tg.relu);
            v.y = (c + 1 < a.Cg) ? nl_apply(x.y, sc[1], sh[1], a.tg.relu) : 0.f;
            v.z = (c + 2 < a.Cg) ? nl_apply(x.z, sc[2], sh[2], a.tg.relu) : 0.f;
            v.w = (c + 3 < a.Cg) ? nl_apply(x.w, sc[3], sh[3], a.tg.relu) : 0.f;
          }
          *reinterpret_cast<float4*>(gl + bv * 32 + cv * 4) = v;
        }
      } else {
        const int cc = tid & 31;
        const int c = cg0 + cc;
        float sc = 0.f, sh = 0.f;
        if (c < a.Cg) nl_coeff(a.tg, n, a.Cg, c, sc, sh);
        for (int bv = tid >> 5; bv < boxvox; bv += 8) {
          const int bx = bv % BX, by = (bv / BX) % BY, bz = bv / (BX * BY);
          const int iz = iz0 + bz, iy = iy0 + by, ix = ix0 + bx;
          float v = 0.f;
          if ((unsigned)iz < (unsigned)a.Dgg && (unsigned)iy < (unsigned)a.Hgg && (unsigned)ix < (unsigned)a.Wgg &&
              c < a.Cg)
            v = nl_apply(gb[iz * a.gsd + iy * a.gsh + ix * a.gsw + c], sc, sh, a.tg.relu);
          gl[bv * 32 + cc] = v;
        }
      }
    }
    // ---- stage D tile: [MT][32 channels cd0..cd0+31]
    {
      const float* db = a.dn + (long long)n * a.dsn;
      const int cc = tid & 31;
      const int c = cd0 + cc;
      float sc = 0.f, sh = 0.f;
      if (c < a.Cd) nl_coeff(a.td, n, a.Cd, c, sc, sh);
      for (int v = tid >> 5; v < MT; v += 8) {
        const int xl = v % TX, yl = (v / TX) % TY, zl = v / (TX * TY);
        const int oz = oz0 + zl, oy = oy0 + yl, ox = ox0 + xl;
        float val = 0.f;
        if (oz < a.Dd && oy < a.Hd && ox < a.Wd && c < a.Cd)
          val = nl_apply(db[oz * a.dsd + oy * a.dsh + ox * a.dsw + c], sc, sh, a.td.relu);
        dl[v * 32 + cc] = val;
      }
    }
    __syncthreads();
    // ---- MFMA: k = voxel pairs
    for (int kk = kstep0; kk < MT / 2; kk += kstride) {
      const int v = 2 * kk + h;
      const int xl = v % TX, yl = (v / TX) % TY, zl = v / (TX * TY);
      const int gidx = ((zl * a.si) * BY + yl * a.si) * BX + xl * a.si;
      const float b = dl[v * 32 + r];
      dbacc += b;
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        if (j < ntw) {
          const float av = gl[(gidx + toff[j]) * 32 + r];
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b, acc[j], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }

  // ---- flush partials
  const int kw = a.ntaps == 1 ? 4 : 1;
  const int sl = a.ntaps == 1 ? (blockIdx.x * 4 + wave) : blockIdx.x;
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    if (j < ntw) {
      const int tap = a.ntaps == 1 ? 0 : wave + 4 * j;
      float* sb = a.slab + (((long long)sl * a.ntaps + tap) * a.CGp + cg0) * a.CDp + cd0 + r;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        sb[(long long)row * a.CDp] = acc[j][i];
      }
    }
  }
  if (a.dbpart != nullptr && blockIdx.y == 0 && (a.ntaps == 1 || wave == 0)) {
    dbacc += __shfl_xor(dbacc, 32, 64);
    if (h == 0) a.dbpart[(long long)sl * a.CDp + cd0 + r] = dbacc;
  }
  (void)kw;
}

// dw[cd][cg][tap] (+)= sum_sl slab[sl][tap][cg][cd]
__global__ void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int nsl, int ntaps, int Cg,
                                    int Cd, int CGp, int CDp, int accumulate) {
  const long long total = (long long)ntaps * Cg * Cd;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int cd = (int)(i % Cd);
    const int cg = (int)((i / Cd) % Cg);
    const int tap = (int)(i / ((long long)Cd * Cg));
    float s = 0.f;
    for (int sl = 0; sl < nsl; ++sl) s += slab[(((long long)sl * ntaps + tap) * CGp + cg) * CDp + cd];
    float* o = dw + ((long long)cd * Cg + cg) * ntaps + tap;
    *o = accumulate ? (*o + s) : s;
  }
}

__global__ void db_reduce_kernel(const float* __restrict__ part, float* __restrict__ db, int nsl, int C, int ld,
                                 int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (int sl = 0; sl < nsl; ++sl) s += part[(long long)sl * ld + c];
  db[c] = accumulate ? db[c] + s : s;
}

// per-channel sum over all voxels of a channels-last tensor, stage 1: part[block][C]
__global__ __launch_bounds__(256) void colsum_kernel(TV x, float* __restrict__ part, long long vox_per_block) {
  // thread = (voxel lane, channel); channel fastest for coalescing; voxel lanes combined through LDS
  __shared__ float red[256];
  const int C = x.c;
  int cpl = 1;
  while (cpl < C && cpl < 256) cpl <<= 1;
  const int nvl = 256 / cpl;
  const int cl = threadIdx.x % cpl, vl = threadIdx.x / cpl;
  const long long nvox = (long long)x.n * x.d * x.h * x.w;
  const long long v0 = blockIdx.x * vox_per_block;
  const long long v1 = (v0 + vox_per_block < nvox) ? v0 + vox_per_block : nvox;
  for (int cb = 0; cb < C; cb += cpl) {
    const int c = cb + cl;
    float s = 0.f;
    if (c < C)
      for (long long v = v0 + vl; v < v1; v += nvl) {
        long long t = v;
        const int xx = (int)(t % x.w); t /= x.w;
        const int yy = (int)(t % x.h); t /= x.h;
        const int zz = (int)(t % x.d);
        const int nn = (int)(t / x.d);
        s += x.p[nn * x.sn + zz * x.sd + yy * x.sh + xx * x.sw + c];
      }
    red[threadIdx.x] = s;
    __syncthreads();
    if (vl == 0 && c < C) {
      float tsum = 0.f;
      for (int j = 0; j < nvl; ++j) tsum += red[j * cpl + cl];
      part[(long long)blockIdx.x * C + c] = tsum;
    }
    __syncthreads();
  }
}

static inline int roundup(int v, int m) { return (v + m - 1) / m * m; }

struct WGeo {
  const mmtta_tensor *g, *dn;
  int si, ntaps, TZ, TY, TX;
  int tz, ty, tx, tiles, S, tps, nsl, CGp, CDp;
  int64_t slab_floats, db_floats, colsum_blocks;
  bool convt;
};

static int wgeometry(const mmtta_conv_desc* d, const mmtta_tensor* x, const mmtta_tensor* dy, WGeo& w) {
  MMTTA_CHECK(d && x && dy && x->ptr && dy->ptr, MMTTA_ERR_INVALID, "wgrad: null argument");
  MMTTA_CHECK(d->op == MMTTA_CONV_FWD || d->op == MMTTA_CONVT_FWD, MMTTA_ERR_INVALID,
              "wgrad: desc.op must name the module (CONV_FWD or CONVT_FWD)");
  MMTTA_CHECK(d->ksize == 1 || d->ksize == 3, MMTTA_ERR_UNSUPPORTED, "wgrad: ksize %d", d->ksize);
  MMTTA_CHECK(d->stride == 1 || d->stride == 2, MMTTA_ERR_UNSUPPORTED, "wgrad: stride %d", d->stride);
  MMTTA_CHECK(d->dtype == MMTTA_F32, MMTTA_ERR_UNSUPPORTED, "wgrad: dtype %d (this build: fp32)", d->dtype);
  MMTTA_CHECK(is_cl(x) && is_cl(dy), MMTTA_ERR_UNSUPPORTED, "wgrad: tensors must be channels-last");
  MMTTA_CHECK(x->c == d->cin && dy->c == d->cout && x->n == dy->n, MMTTA_ERR_INVALID, "wgrad: channel/batch mismatch");
  w.convt = d->op == MMTTA_CONVT_FWD;
  if (w.convt) MMTTA_CHECK(d->ksize == 3 && d->stride == 2, MMTTA_ERR_UNSUPPORTED, "wgrad: conv_transpose is k3 s2 only");
  w.g = w.convt ? dy : x;
  w.dn = w.convt ? x : dy;
  w.si = d->stride;
  w.ntaps = d->ksize * d->ksize * d->ksize;
  const int gd[3] = {w.g->d, w.g->h, w.g->w}, dd[3] = {w.dn->d, w.dn->h, w.dn->w};
  for (int i = 0; i < 3; ++i) {
    const int want = w.convt ? gd[i] / 2 : (d->stride == 1 ? gd[i] : (gd[i] + 1) / 2);
    MMTTA_CHECK(dd[i] == want && (!w.convt || gd[i] % 2 == 0), MMTTA_ERR_INVALID,
                "wgrad: spatial mismatch on axis %d (fine %d, coarse %d)", i, gd[i], dd[i]);
  }
  if (w.si == 1) { w.TZ = 4; w.TY = 4; w.TX = 8; } else { w.TZ = 2; w.TY = 2; w.TX = 8; }
  w.tz = (w.dn->d + w.TZ - 1) / w.TZ;
  w.ty = (w.dn->h + w.TY - 1) / w.TY;
  w.tx = (w.dn->w + w.TX - 1) / w.TX;
  w.tiles = w.tz * w.ty * w.tx * x->n;
  w.CGp = roundup(w.g->c, 32);
  w.CDp = roundup(w.dn->c, 32);
  const int blocks_cc = (w.CGp / 32) * (w.CDp / 32);
  int S = 512 / blocks_cc;
  if (S < 1) S = 1;
  if (S > w.tiles) S = w.tiles;
  w.tps = (w.tiles + S - 1) / S;
  w.S = (w.tiles + w.tps - 1) / w.tps;
  w.nsl = w.S * (w.ntaps == 1 ? 4 : 1);
  w.slab_floats = (int64_t)w.nsl * w.ntaps * w.CGp * w.CDp;
  w.colsum_blocks = 0;
  if (w.convt) {
    const int64_t nvox = (int64_t)dy->n * dy->d * dy->h * dy->w;
    w.colsum_blocks = nvox < 1024 ? 1 : (nvox + 1023) / 1024;
    if (w.colsum_blocks > 2048) w.colsum_blocks = 2048;
    w.db_floats = w.colsum_blocks * dy->c;
  } else {
    w.db_floats = (int64_t)w.nsl * w.CDp;
  }
  return MMTTA_OK;
}

template <int TZ, int TY, int TX>
static int launch_wgrad(const WArgs& a, int S, hipStream_t s) {
  const int ext = a.ntaps == 1 ? 0 : 2;
  const int BZ = (TZ - 1) * a.si + ext + 1, BY = (TY - 1) * a.si + ext + 1, BX = (TX - 1) * a.si + ext + 1;
  const size_t lds = ((size_t)BZ * BY * BX + TZ * TY * TX) * 32 * sizeof(float);
  auto kern = wgrad_f32_kernel<TZ, TY, TX>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  dim3 grid(S, a.CGp / 32, a.CDp / 32);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
  return launch_status("conv wgrad");
}

}  // namespace mmtta

using namespace mmtta;

extern "C" int64_t mmtta_conv_wgrad_workspace_bytes(const mmtta_conv_desc* d, const mmtta_tensor* x,
                                                    const mmtta_tensor* dy) {
  WGeo w;
  if (wgeometry(d, x, dy, w)) return -1;
  return (w.slab_floats + w.db_floats) * (int64_t)sizeof(float);
}

extern "C" int mmtta_conv_wgrad(const mmtta_conv_desc* d, const mmtta_tensor* x, const mmtta_norm_on_load* x_norm,
                                const mmtta_tensor* dy, float* dw, float* db, int accumulate, void* workspace,
                                int64_t workspace_bytes, void* stream) {
  WGeo w;
  int st = wgeometry(d, x, dy, w);
  if (st) return st;
  MMTTA_CHECK(dw != nullptr, MMTTA_ERR_INVALID, "wgrad: null dw");
  const int64_t need = (w.slab_floats + w.db_floats) * 4;
  MMTTA_CHECK(workspace != nullptr && workspace_bytes >= need, MMTTA_ERR_WORKSPACE, "wgrad: workspace %lld bytes, need %lld",
              (long long)workspace_bytes, (long long)need);
  hipStream_t s = (hipStream_t)stream;
  WArgs a;
  a.g = (const float*)w.g->ptr; a.gsn = w.g->sn; a.gsd = w.g->sd; a.gsh = w.g->sh; a.gsw = w.g->sw;
  a.Cg = w.g->c; a.Dgg = w.g->d; a.Hgg = w.g->h; a.Wgg = w.g->w;
  a.dn = (const float*)w.dn->ptr; a.dsn = w.dn->sn; a.dsd = w.dn->sd; a.dsh = w.dn->sh; a.dsw = w.dn->sw;
  a.Cd = w.dn->c; a.Dd = w.dn->d; a.Hd = w.dn->h; a.Wd = w.dn->w;
  // the module input carries the norm-on-load; the gradient tensor is read as is
  a.tg = w.convt ? nl(nullptr) : nl(x_norm);
  a.td = w.convt ? nl(x_norm) : nl(nullptr);
  a.N = x->n;
  a.si = w.si; a.ntaps = w.ntaps;
  a.slab = (float*)workspace;
  float* dbws = (float*)workspace + w.slab_floats;
  a.dbpart = (db != nullptr && !w.convt) ? dbws : nullptr;
  a.tz = w.tz; a.ty = w.ty; a.tx = w.tx; a.tiles = w.tiles; a.tiles_per_split = w.tps;
  a.CGp = w.CGp; a.CDp = w.CDp;
  a.gvec4 = ((((uintptr_t)w.g->ptr) % 16 == 0) && w.g->sw % 4 == 0 && w.g->sh % 4 == 0 && w.g->sd % 4 == 0 &&
             w.g->sn % 4 == 0) ? 1 : 0;
  a.dvec4 = 0;
  st = (w.si == 1) ? launch_wgrad<4, 4, 8>(a, w.S, s) : launch_wgrad<2, 2, 8>(a, w.S, s);
  if (st) return st;
  {
    const long long total = (long long)w.ntaps * a.Cg * a.Cd;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, s, a.slab, dw, w.nsl, w.ntaps, a.Cg, a.Cd, w.CGp,
                       w.CDp, accumulate);
    st = launch_status("wgrad reduce");
    if (st) return st;
  }
  if (db != nullptr) {
    if (!w.convt) {
      hipLaunchKernelGGL(db_reduce_kernel, dim3((a.Cd + 63) / 64), dim3(64), 0, s, dbws, db, w.nsl, a.Cd, w.CDp, accumulate);
    } else {
      const int64_t nvox = (int64_t)dy->n * dy->d * dy->h * dy->w;
      const long long vpb = (nvox + w.colsum_blocks - 1) / w.colsum_blocks;
      hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)w.colsum_blocks), dim3(256), 0, s, tv(dy), dbws, vpb);
      st = launch_status("bias colsum");
      if (st) return st;
      hipLaunchKernelGGL(db_reduce_kernel, dim3((dy->c + 63) / 64), dim3(64), 0, s, dbws, db, (int)w.colsum_blocks, dy->c,
                         dy->c, accumulate);
    }
    st = launch_status("bias reduce");
  }
  return st;
}
