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
static inline int roundup(int v, int m) { return (v + m - 1) / m * m; }

struct WArgs {
  const float* g; long long gsn, gsd, gsh, gsw; int Cg, Dgg, Hgg, Wgg; NL tg;
  const float* dn; long long dsn, dsd, dsh, dsw; int Cd, Dd, Hd, Wd; NL td;
  int N;
  int si, ntaps;
  float* slab;     // [nsl][ntaps][CGp][CDp]
  float* dbpart;   // [nsl][CDp] or null
  int tz, ty, tx, tiles, tiles_per_split;
  int S, tiles_set;   // slabs and tiles per parameter set (one set: S = launch slabs, tiles_set = tiles)
  int CGp, CDp;
  int gvec4, dvec4;
  int g_bf, d_bf;   // storage of the gathered / dense tensor: 1 = bf16 elements (the forward activation of bf16 precision)
  int convt;        // transposed module: the dense operand is the module input (carries the norm-on-load)
};

// NTW = accumulators per wave: 7 for the 27-tap kernel (taps wave, wave+4, ...; slot 27 is a dummy that is
// never flushed), 1 for a single tap (the four waves then split the voxel pairs instead).
template <int TZ, int TY, int TX, int NTW, bool GBF = false, bool DBF = false>
__global__ __launch_bounds__(256) void wgrad_f32_kernel(WArgs a) {
  extern __shared__ float lds[];
  constexpr int MT = TZ * TY * TX;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: keeps every tap branch uniform
  const int h = lane >> 5, r = lane & 31;
  const int ext = NTW == 1 ? 0 : 2;
  const int BZ = (TZ - 1) * a.si + ext + 1, BY = (TY - 1) * a.si + ext + 1, BX = (TX - 1) * a.si + ext + 1;
  const int boxvox = BZ * BY * BX;
  const int dmin = NTW == 1 ? 0 : -1;
  float* gl = lds;                 // [boxvox][32]
  float* dl = lds + boxvox * 32;   // [MT][32]
  const int cg0 = blockIdx.y * 32, cd0 = blockIdx.z * 32;

  int toff[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    const int tap = NTW == 1 ? 0 : min(wave + 4 * j, 26);
    toff[j] = NTW == 1 ? 0 : (((tap / 9) * BY + ((tap / 3) % 3)) * BX + (tap % 3)) * 32;
  }
  const int kstep0 = NTW == 1 ? wave : 0, kstride = NTW == 1 ? 4 : 1;

  f32x16 acc[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
  float dbacc = 0.f;

  // slab index = blockIdx.x; the TILE RANGE of a slab follows the XCD-contiguous order, so the workgroups of one XCD
  // sweep one contiguous part of the volume and re-read each other's halo rows from their own L2
  // slab sx = slab (sx % S) of parameter set (sx / S): a set's slabs cover ITS batch items only, in the order a launch of
  // those items alone would use (the reduce kernels then sum a set's rows in that same order)
  const int sx = (int)xcd_contiguous_id(blockIdx.x, gridDim.x);
  const int qset = sx / a.S, sls = sx - qset * a.S;
  const int t0 = qset * a.tiles_set + sls * a.tiles_per_split;
  const int t1 = min((qset + 1) * a.tiles_set, t0 + a.tiles_per_split);
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
      const float* gb = a.g;
      const long long gbo = (long long)n * a.gsn;
      if (a.gvec4) {
        const int cv = tid & 7;
        const int c = cg0 + cv * 4;
        float sc[4], sh[4];
        nl_coeff_vec<4>(a.tg, n, a.Cg, c, sc, sh);
        constexpr int U = 4;       // 4 box voxels per trip: loads first, then transform + LDS store
        for (int bv0 = tid >> 3; bv0 < boxvox; bv0 += 32 * U) {
          float4 xin[U];
          bool ok[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int bv = bv0 + 32 * u;
            const int bx = bv % BX, by = (bv / BX) % BY, bz = bv / (BX * BY);
            const int iz = iz0 + bz, iy = iy0 + by, ix = ix0 + bx;
            ok[u] = bv < boxvox && (unsigned)iz < (unsigned)a.Dgg && (unsigned)iy < (unsigned)a.Hgg &&
                    (unsigned)ix < (unsigned)a.Wgg && c < a.Cg;
            xin[u] = ld4_t<GBF>(gb, gbo + (long long)min(max(iz, 0), a.Dgg - 1) * a.gsd + (long long)min(max(iy, 0), a.Hgg - 1) * a.gsh +
                                      (long long)min(max(ix, 0), a.Wgg - 1) * a.gsw + min(c, (a.Cg - 1) & ~3));
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int bv = bv0 + 32 * u;
            if (bv < boxvox) {
              float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
              if (ok[u]) {
                v.x = nl_apply(xin[u].x, sc[0], sh[0], a.tg.relu);
                v.y = (c + 1 < a.Cg) ? nl_apply(xin[u].y, sc[1], sh[1], a.tg.relu) : 0.f;
                v.z = (c + 2 < a.Cg) ? nl_apply(xin[u].z, sc[2], sh[2], a.tg.relu) : 0.f;
                v.w = (c + 3 < a.Cg) ? nl_apply(xin[u].w, sc[3], sh[3], a.tg.relu) : 0.f;
              }
              *reinterpret_cast<float4*>(gl + bv * 32 + cv * 4) = v;
            }
          }
        }
      } else {
        const int cc = tid & 31;
        const int c = cg0 + cc;
        float sc = 0.f, sh = 0.f;
        nl_coeff_vec<1>(a.tg, n, a.Cg, c, &sc, &sh);
        for (int bv = tid >> 5; bv < boxvox; bv += 8) {
          const int bx = bv % BX, by = (bv / BX) % BY, bz = bv / (BX * BY);
          const int iz = iz0 + bz, iy = iy0 + by, ix = ix0 + bx;
          float v = 0.f;
          if ((unsigned)iz < (unsigned)a.Dgg && (unsigned)iy < (unsigned)a.Hgg && (unsigned)ix < (unsigned)a.Wgg &&
              c < a.Cg)
            v = nl_apply(ld1_t<GBF>(gb, gbo + (long long)iz * a.gsd + (long long)iy * a.gsh + (long long)ix * a.gsw + c), sc, sh, a.tg.relu);
          gl[bv * 32 + cc] = v;
        }
      }
    }
    // ---- stage D tile: [MT][32 channels cd0..cd0+31]; every load unconditional from a clamped address, all of a
    // thread's loads in flight together (a load behind a per-item branch is waited for at once: 16 exposed round
    // trips per tile before)
    {
      const float* db = a.dn;
      const long long dbo = (long long)n * a.dsn;
      if (a.dvec4) {
        const int cv = tid & 7;
        const int c = cd0 + cv * 4;
        float sc[4], sh[4];
        nl_coeff_vec<4>(a.td, n, a.Cd, c, sc, sh);
        const int cl4 = min(c, (a.Cd - 1) & ~3);
        constexpr int NQ = MT / 32;
        float4 raw[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int v = (tid >> 3) + 32 * q;
          const int xl = v % TX, yl = (v / TX) % TY, zl = v / (TX * TY);
          raw[q] = ld4_t<DBF>(db, dbo + (long long)min(oz0 + zl, a.Dd - 1) * a.dsd + (long long)min(oy0 + yl, a.Hd - 1) * a.dsh +
                                    (long long)min(ox0 + xl, a.Wd - 1) * a.dsw + cl4);
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int v = (tid >> 3) + 32 * q;
          const int xl = v % TX, yl = (v / TX) % TY, zl = v / (TX * TY);
          const bool ok = oz0 + zl < a.Dd && oy0 + yl < a.Hd && ox0 + xl < a.Wd;
          float4 o;
          o.x = (ok && c < a.Cd) ? nl_apply(raw[q].x, sc[0], sh[0], a.td.relu) : 0.f;
          o.y = (ok && c + 1 < a.Cd) ? nl_apply(raw[q].y, sc[1], sh[1], a.td.relu) : 0.f;
          o.z = (ok && c + 2 < a.Cd) ? nl_apply(raw[q].z, sc[2], sh[2], a.td.relu) : 0.f;
          o.w = (ok && c + 3 < a.Cd) ? nl_apply(raw[q].w, sc[3], sh[3], a.td.relu) : 0.f;
          *reinterpret_cast<float4*>(dl + v * 32 + cv * 4) = o;
        }
      } else {
        const int cc = tid & 31;
        const int c = cd0 + cc;
        float sc = 0.f, sh = 0.f;
        nl_coeff_vec<1>(a.td, n, a.Cd, c, &sc, &sh);
        const int cl = min(c, a.Cd - 1);
        constexpr int NQ = MT / 8;
        float raw[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int v = (tid >> 5) + 8 * q;
          const int xl = v % TX, yl = (v / TX) % TY, zl = v / (TX * TY);
          raw[q] = ld1_t<DBF>(db, dbo + (long long)min(oz0 + zl, a.Dd - 1) * a.dsd + (long long)min(oy0 + yl, a.Hd - 1) * a.dsh +
                                    (long long)min(ox0 + xl, a.Wd - 1) * a.dsw + cl);
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int v = (tid >> 5) + 8 * q;
          const int xl = v % TX, yl = (v / TX) % TY, zl = v / (TX * TY);
          const bool ok = oz0 + zl < a.Dd && oy0 + yl < a.Hd && ox0 + xl < a.Wd && c < a.Cd;
          dl[v * 32 + cc] = ok ? nl_apply(raw[q], sc, sh, a.td.relu) : 0.f;
        }
      }
    }
    __syncthreads();
    // ---- MFMA: k = voxel pairs; straight-line body (reads of a step are independent of its MFMAs)
#pragma unroll 2
    for (int kk = kstep0; kk < MT / 2; kk += kstride) {
      const int v = 2 * kk + h;
      const int xl = v % TX, yl = (v / TX) % TY, zl = v / (TX * TY);
      const int gaddr = (((zl * a.si) * BY + yl * a.si) * BX + xl * a.si) * 32 + r;
      const float b = dl[v * 32 + r];
      float av[NTW];
#pragma unroll
      for (int j = 0; j < NTW; ++j) av[j] = gl[gaddr + toff[j]];
      dbacc += b;
#pragma unroll
      for (int j = 0; j < NTW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], b, acc[j], 0, 0, 0);
    }
    __syncthreads();
  }

  // ---- flush partials
  const int sl = NTW == 1 ? (sx * 4 + wave) : sx;
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    const int tap = NTW == 1 ? 0 : wave + 4 * j;
    if (tap < a.ntaps) {
      float* sb = a.slab + (((long long)sl * a.ntaps + tap) * a.CGp + cg0) * a.CDp + cd0 + r;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        sb[(long long)row * a.CDp] = acc[j][i];
      }
    }
  }
  if (a.dbpart != nullptr && blockIdx.y == 0 && (NTW == 1 || wave == 0)) {
    dbacc += __shfl_xor(dbacc, 32, 64);
    if (h == 0) a.dbpart[(long long)sl * a.CDp + cd0 + r] = dbacc;
  }
}

// dw[cd][cg][tap] (+)= sum_sl slab[sl][tap][cg][cd]   (torch weight layout out of the [tap][cg][cd] slabs)
// The transpose goes through LDS so that both sides stay coalesced: reads run along cd, writes are runs of
// `ntaps` consecutive floats per (cd, cg).  27 taps: one block = one cg x 64 cd;  1 tap: 32 cg x 32 cd.
__global__ __launch_bounds__(256) void wgrad_reduce27_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                             int nsl, int Cg, int Cd, int CGp, int CDp, int accumulate,
                                                             const float* __restrict__ dbpart, float* __restrict__ db,
                                                             int db_nsl, PSets ps) {
  __shared__ float tile[32][28];
  {  // parameter set blockIdx.z: its rows of the slab / bias partials, its dw / db
    const int q = blockIdx.z;
    slab += (long long)q * nsl * 27 * CGp * CDp;
    dw += pset_weight_elems(ps, q);
    if (db != nullptr) { dbpart += (long long)q * db_nsl * CDp; db += pset_bias_elems(ps, q); }
  }
  const int cd0 = blockIdx.y * 32;
  if ((int)blockIdx.x == Cg) {
    // bias gradient rows ride in the same launch: db[cd] (+)= sum_sl dbpart[sl][cd]; 8 threads per channel
    const int cdl = threadIdx.x & 31, part = threadIdx.x >> 5;
    const int cd = min(cd0 + cdl, Cd - 1);
    float s4[4] = {0.f, 0.f, 0.f, 0.f};
    for (int sl = part; sl < db_nsl; sl += 32) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int q = sl + 8 * u;
        const float v = dbpart[(long long)min(q, db_nsl - 1) * CDp + cd];
        s4[u] += q < db_nsl ? v : 0.f;
      }
    }
    float* red = &tile[0][0];
    red[threadIdx.x] = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    __syncthreads();
    if (threadIdx.x < 32 && cd0 + (int)threadIdx.x < Cd) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) t += red[q * 32 + threadIdx.x];
      float* o = db + cd0 + threadIdx.x;
      *o = accumulate ? (*o + t) : t;
    }
    return;
  }
  // one block = one cg x 32 cd x 27 taps = 864 sums: up to 4 per thread, all advancing together through the slabs
  // (4 slabs x 4 items = 16 independent loads per trip, unconditional from clamped addresses)
  const int cg = blockIdx.x;
  const long long st = (long long)27 * CGp * CDp;
  const float* pit[4];
  bool live[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int i = threadIdx.x + 256 * it;
    const int cdl = i & 31, tap = min(i >> 5, 26);
    live[it] = i < 27 * 32 && cd0 + cdl < Cd;
    pit[it] = slab + ((long long)tap * CGp + cg) * CDp + min(cd0 + cdl, CDp - 1);
  }
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int sl = 0; sl < nsl; sl += 4) {
    float v[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int it = 0; it < 4; ++it) v[it][u] = pit[it][(long long)min(sl + u, nsl - 1) * st];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int it = 0; it < 4; ++it) acc[it] += sl + u < nsl ? v[it][u] : 0.f;
  }
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int i = threadIdx.x + 256 * it;
    if (i < 27 * 32) tile[i & 31][i >> 5] = live[it] ? acc[it] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 27 * 32; i += 256) {
    const int cdl = i / 27, tap = i % 27;
    const int cd = cd0 + cdl;
    if (cd < Cd) {
      const long long idx = ((long long)cd * Cg + cg) * 27 + tap;
      const float v = tile[cdl][tap];
      dw[idx] = accumulate ? (dw[idx] + v) : v;
    }
  }
}

__global__ __launch_bounds__(256) void wgrad_reduce1_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                            int nsl, int Cg, int Cd, int CGp, int CDp, int accumulate, PSets ps) {
  __shared__ float tile[32][33];
  slab += (long long)blockIdx.z * nsl * CGp * CDp;       // parameter set blockIdx.z
  dw += pset_weight_elems(ps, blockIdx.z);
  const int cg0 = blockIdx.x * 32, cd0 = blockIdx.y * 32;
  for (int i = threadIdx.x; i < 32 * 32; i += 256) {
    const int cdl = i & 31, cgl = i >> 5;
    float s = 0.f;
    if (cg0 + cgl < Cg && cd0 + cdl < Cd)
      for (int sl = 0; sl < nsl; ++sl) s += slab[((long long)sl * CGp + cg0 + cgl) * CDp + cd0 + cdl];
    tile[cdl][cgl] = s;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 32 * 32; i += 256) {
    const int cgl = i & 31, cdl = i >> 5;
    if (cg0 + cgl < Cg && cd0 + cdl < Cd) {
      const long long idx = (long long)(cd0 + cdl) * Cg + cg0 + cgl;
      const float v = tile[cdl][cgl];
      dw[idx] = accumulate ? (dw[idx] + v) : v;
    }
  }
}

// first reduction stage when many spatial splits exist: out[chunk][e] = sum of 32 consecutive slabs (coalesced)
__global__ __launch_bounds__(256) void slab_prereduce_kernel(const float* __restrict__ slab, float* __restrict__ out,
                                                             int nsl, long long elems) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= elems) return;
  slab += (long long)blockIdx.z * nsl * elems;           // parameter set blockIdx.z: its slabs, its chunks
  out += (long long)blockIdx.z * gridDim.y * elems;
  const int s0 = blockIdx.y * 32, s1 = min(nsl, s0 + 32);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int sl = s0; sl < s1; sl += 8) {          // 8 independent loads per trip (clamped, masked)
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = slab[(long long)min(sl + u, s1 - 1) * elems + e];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] += sl + u < s1 ? v[u] : 0.f;
  }
  out[(long long)blockIdx.y * elems + e] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
}

// db[c] (+)= sum_sl part[sl*ld + c]: one wave per channel, lanes stride the partial rows
__global__ __launch_bounds__(64) void db_reduce_kernel(const float* __restrict__ part, float* __restrict__ db, int nsl,
                                                       int C, int ld, int accumulate, PSets ps, int is_weight) {
  const int c = blockIdx.x;
  part += (long long)blockIdx.y * nsl * ld;              // parameter set blockIdx.y
  db += is_weight ? pset_weight_elems(ps, blockIdx.y) : pset_bias_elems(ps, blockIdx.y);
  float s = 0.f;
  for (int sl = threadIdx.x; sl < nsl; sl += 64) s += part[(long long)sl * ld + c];
  s = wave_sum(s);
  if (threadIdx.x == 0) db[c] = accumulate ? db[c] + s : s;
}

// ------------------------------------------------------------------ bf16 helpers of the matrix-core weight gradients
typedef __bf16 wbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 wbf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned int wpack2(float lo, float hi) {
  wbf16x2 v; v[0] = (__bf16)lo; v[1] = (__bf16)hi;
  return __builtin_bit_cast(unsigned int, v);
}


// ------------------------------------------------------------------ transposed-read weight gradient (bf16 operands)
// The kernels above keep the contraction index (voxels) contiguous per lane by transposing while staging: a thread owns
// channels, walks x with element (or 4-channel) loads and writes three x-shifted [channel][voxel] copies.  gfx950 can do
// that transpose in the LDS read instead (ds_read_b64_tr_b16: a 16-lane group reads 4 rows x 16 columns of 16-bit
// elements and each lane receives one column), so here both operands sit in LDS exactly as they sit in HBM -
// [voxel][32 channels] bf16, 64-byte rows - staged with 16-byte (bf16 storage: 8-byte) coalesced loads, ONE copy of the
// gathered box, and a tap is a constant byte offset.  Lane l of the MFMA operand (column l & 31, k = 8 (l >> 5) + j)
// receives, from two transposed reads, channel l & 31 of the 8 voxels x = 0..7 of tile row 2 ks + (l >> 5): the k order
// is the same for A and B, which is all the contraction needs.  Every address in the k loop is lane base + immediate.
// Stride 2: the box's x positions are stored even-first so that the stride-2 walk of a tap is contiguous (4 rows of a
// transposed read 64 bytes apart: conflict-free), the same as stride 1.
template <int TZ, int TY, int SI>
struct WTGeo {
  static constexpr int BZ = (TZ - 1) * SI + 3, BY = (TY - 1) * SI + 3, BX = 7 * SI + 3;
  static constexpr int XH = (BX + 1) / 2;
  static constexpr int NG = BZ * BY * BX, ND = TZ * TY * 8;
  static constexpr int G_BYTES = NG * 64, D_BYTES = ND * 64;
  static constexpr int LDS_BYTES = G_BYTES + D_BYTES + 256;       // + scale / shift of the 32 module-input channels
  __host__ __device__ static constexpr int lds_bytes(int nb) { return G_BYTES + nb * (D_BYTES + 256); }   // nb dense column blocks
  static constexpr int NK = ND / 16;                  // k steps: two tile rows of 8 voxels each
  // staging: an item = 8 channels (16 bytes of the bf16 image) of one voxel; a thread keeps ONE (x, channel chunk) of the
  // box and walks the box rows RPP at a time, so x, the channel offset and the LDS column are per-thread constants
  static constexpr int IPR = BX * 4;                  // items per box row
  static constexpr int RPP = 256 / IPR;               // box rows per pass (threads >= RPP * IPR repeat row RPP - 1)
  static constexpr int NROW = BZ * BY;
  static constexpr int GP = NROW / RPP;               // passes over the box
  static constexpr int DP = ND * 4 / 256;             // dense-tile items per thread
  static_assert(NROW % RPP == 0 && (ND * 4) % 256 == 0 && RPP <= BY, "whole passes; at most one y wrap inside a pass");
  __host__ __device__ static constexpr int xmap(int x) { return SI == 1 ? x : (x & 1) * XH + (x >> 1); }
  __host__ __device__ static constexpr int tap_off(int tap) {        // bytes; tap = (kz*3 + ky)*3 + kx
    return (((tap / 9) * BY + (tap / 3) % 3) * BX + xmap(tap % 3)) * 64;
  }
  __host__ __device__ static constexpr int row_off(int ks) {         // bytes; tile row 2 ks (the lane adds its half)
    return ((((2 * ks) / TY) * SI * BY + ((2 * ks) % TY) * SI) * BX) * 64;
  }
};

typedef short wshort4 __attribute__((ext_vector_type(4)));
typedef short wshort8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) wshort4* wlds4_t;

// 8 k values of one operand column: two transposed reads 4 voxels (256 bytes) apart
#define MMTTA_TR_FRAG(ptr, off)                                                                          \
  __builtin_bit_cast(wbf16x8, __builtin_shufflevector(__builtin_amdgcn_ds_read_tr16_b64_v4i16((wlds4_t)((ptr) + (off))),       \
                                                      __builtin_amdgcn_ds_read_tr16_b64_v4i16((wlds4_t)((ptr) + (off) + 256)), \
                                                      0, 1, 2, 3, 4, 5, 6, 7))

// Staging is the vector-ALU cost of this kernel, and with several volumes in flight vector-ALU issue is what bounds the
// chip (profiles/r02c_sq_counters.md), so the loader is built for few instructions per element:
//  * an item is 8 channels; a thread owns one (x, chunk) column of the box: x bounds, x / channel offsets, the x
//    de-interleave of stride 2 and the LDS column are computed once per tile, the LDS row is an immediate offset;
//  * a pass covers RPP whole box rows: z is wave-uniform up to one wrap (two scalar candidates, one select), y is one add;
//  * element offsets are 32-bit, products are 24-bit multiply-adds (strides < 2^24, tensor < 2^31 elements: host check);
//  * the norm-on-load exists on the module-input side only (template TD: the dense operand of a transposed module, else
//    the gathered one): the gradient side is a plain fp32 -> bf16 pack;  ReLU is max(x, lo) with lo = 0 or -inf.
// Software pipeline over the tiles of a workgroup: the dense tile and the first PG box passes of tile i+1 are requested
// right before tile i's MFMA loop (which reads LDS only) and land under it; the remaining passes are requested first
// thing in tile i+1 and land under the commit of the prefetched ones.
// NB = 2: ONE workgroup multiplies the gathered box with two 32-channel blocks of the dense operand (14 accumulator blocks:
// one workgroup per CU).  The stride-2 layers stage a box of 12x the tile's voxels for 28 MFMAs a wave - the staging (vector
// ALU, L2) is their cost, and every dense column block used to repeat it (conv 32->64 at 64^3: 2x, convT 128->32: 4x).
template <int TZ, int TY, int SI, bool GBF, bool DBF, bool TD, int NB = 1>
__global__ __launch_bounds__(256, (NB == 1 && SI == 1) ? 2 : 1) void wgrad_tr_kernel(WArgs a) {
  using G = WTGeo<TZ, TY, SI>;
  static_assert(TY % 2 == 0, "a k step is two rows of one z slice");
  extern __shared__ float lds[];
  unsigned char* gl = reinterpret_cast<unsigned char*>(lds);
  unsigned char* dl = gl + G::G_BYTES;                  // NB dense images back to back
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, r = lane & 31;
  const int cg0 = blockIdx.y * 32, cd0 = blockIdx.z * 32 * NB;

  // operand addresses: lane part (group row q, 8-byte column chunk, row half) + tap (A only)
  const int lq = (lane & 15) >> 2, lp = lane & 3, lc = (lane >> 4) & 1;
  const unsigned char* dread = dl + (h * 8 + lq) * 64 + lc * 32 + lp * 8;
  const unsigned char* gread[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    const int tap = min(wave + 4 * j, 26);
    gread[j] = gl + (h * SI * G::BX + lq) * 64 + lc * 32 + lp * 8 + G::tap_off(tap);
  }
  f32x16 acc[NB][7];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int j = 0; j < 7; ++j)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[nb][j][i] = 0.f;
  float dbs[NB][8];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int c = 0; c < 8; ++c) dbs[nb][c] = 0.f;
  const bool want_db = a.dbpart != nullptr && blockIdx.y == 0;

  // ---- staging roles (per-thread constants)
  const int rsub = min(tid / G::IPR, G::RPP - 1);                // box row inside a pass
  const int rem = tid % G::IPR, bx = rem >> 2, c8 = rem & 3;     // box x, 8-channel chunk
  const int bxm = SI == 1 ? bx : (bx & 1) * G::XH + (bx >> 1);
  unsigned char* gst = gl + (rsub * G::BX + bxm) * 64 + c8 * 16;  // + pass * RPP * BX * 64
  const int dvox0 = tid >> 2, d8 = tid & 3;                      // dense item p: voxel dvox0 + 64 p, chunk d8
  unsigned char* dst = dl + dvox0 * 64 + d8 * 16;                // + p * 4096
  const int dzl = dvox0 / (TY * 8), dyl = (dvox0 >> 3) % TY, dxl = dvox0 & 7;     // (pass p adds 64 / (TY * 8) to z)
  const int gcb = cg0 + 8 * c8, dcb = cd0 + 8 * d8;               // (dense column block nb: + 32 nb)
  // clamped (always valid) load channels; rows of bf16 tensors are padded to 8 channels, of fp32 tensors to 4
  const unsigned gc_lo = GBF ? min(gcb, (a.Cg - 1) & ~7) : min(gcb, (a.Cg - 1) & ~3), gc_hi = min(gcb + 4, (a.Cg - 1) & ~3);
  const unsigned dc_lo = DBF ? min(dcb, (a.Cd - 1) & ~7) : min(dcb, (a.Cd - 1) & ~3), dc_hi = min(dcb + 4, (a.Cd - 1) & ~3);
  // (column block 1 of a pair is whole: the host pairs blocks only when CDp / 32 is even and rows are padded to 8 channels)
  const unsigned dc_lo1 = DBF ? min(dcb + 32, (a.Cd - 1) & ~7) : min(dcb + 32, (a.Cd - 1) & ~3), dc_hi1 = min(dcb + 36, (a.Cd - 1) & ~3);
  const bool gtail = (a.Cg & 7) != 0, dtail = (a.Cd & 7) != 0;  // only then can a chunk hold channels past the last one
  auto pair_mask = [](int c, int C) { return (c < C ? 0xffffu : 0u) | (c + 1 < C ? 0xffff0000u : 0u); };
  const unsigned gsd = (unsigned)a.gsd, gsh = (unsigned)a.gsh, gsw = (unsigned)a.gsw;
  const unsigned dsd = (unsigned)a.dsd, dsh = (unsigned)a.dsh, dsw = (unsigned)a.dsw;

  // slab sx = slab (sx % S) of parameter set (sx / S): a set's slabs cover ITS batch items only, in the order a launch of
  // those items alone would use (the reduce kernels then sum a set's rows in that same order)
  const int sx = (int)xcd_contiguous_id(blockIdx.x, gridDim.x);
  const int qset = sx / a.S, sls = sx - qset * a.S;
  const int t0 = qset * a.tiles_set + sls * a.tiles_per_split;
  const int t1 = min((qset + 1) * a.tiles_set, t0 + a.tiles_per_split);
  const int tpn = a.tz * a.ty * a.tx;
  constexpr int GP = G::GP, DP = G::DP;
  constexpr int PREF = (GBF || DBF) ? 40 : 32;                         // registers the cross-loop prefetch may hold
  constexpr int PGmax = (PREF - NB * DP * (DBF ? 4 : 8)) / (GBF ? 4 : 8);
  constexpr int PG = PGmax < 0 ? 0 : (PGmax < GP ? PGmax : GP);        // box passes prefetched across the MFMA loop
  Oct8<GBF> gv[GP];
  Oct8<DBF> dq[NB * DP];
  unsigned pok = 0u;                             // bit p: box item of pass p lies inside the tensor
  unsigned dok = 0u;                             // bit p: dense item p lies inside the tensor
  int pn = -1, poz0 = 0, poy0 = 0, pox0 = 0;     // tile whose loads are in gv[0..PG) / dq
  int cn = -1;                                   // batch item the transform coefficients belong to
  float* coefl = reinterpret_cast<float*>(dl + NB * G::D_BYTES);     // [NB][2][32]: scale, shift of the module-input channels
  const float relu_lo = (TD ? a.td.relu : a.tg.relu) ? 0.f : -__builtin_inff();
  unsigned gxoff = 0u;                           // per tile: x / channel part of the box offsets, x in bounds
  bool gxok = false;

  auto tile_x = [&](int ix0) {
    const int ix = ix0 + bx;
    gxok = (unsigned)ix < (unsigned)a.Wgg;
    gxoff = __umul24((unsigned)min(max(ix, 0), a.Wgg - 1), gsw);
  };
  auto load_g = [&](auto pc, const float* gbase, int iz0, int iy0) {
    constexpr int P = decltype(pc)::value;
    constexpr int row0 = G::RPP * P, bz0 = row0 / G::BY, by0 = row0 % G::BY;
    constexpr bool can_wrap = by0 + G::RPP > G::BY;
    int by = by0 + rsub;
    bool wrap = false;
    if constexpr (can_wrap) {
      wrap = by >= G::BY;
      by = wrap ? by - G::BY : by;
    }
    const int izA = iz0 + bz0, izB = izA + 1;                    // wave-uniform candidates
    const unsigned zoA = __umul24((unsigned)min(max(izA, 0), a.Dgg - 1), gsd), zoB = __umul24((unsigned)min(max(izB, 0), a.Dgg - 1), gsd);
    const bool zkA = (unsigned)izA < (unsigned)a.Dgg, zkB = (unsigned)izB < (unsigned)a.Dgg;
    const unsigned zo = (can_wrap && wrap) ? zoB : zoA;
    const bool zk = (can_wrap && wrap) ? zkB : zkA;
    const int iy = iy0 + by;
    const bool ok = zk && gxok && (unsigned)iy < (unsigned)a.Hgg;
    pok |= (ok ? 1u : 0u) << P;
    const unsigned off = zo + __umul24((unsigned)min(max(iy, 0), a.Hgg - 1), gsh) + gxoff;
    gv[P] = oct8_ld<GBF>(gbase, off + gc_lo, off + gc_hi);
  };
  auto commit_g = [&](auto pc, const float (&sc)[8], const float (&sh)[8]) {
    constexpr int P = decltype(pc)::value;
    float v[8];
    oct8_f8(gv[P], v);
    if constexpr (!TD) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = fmaxf(fmaf(v[j], sc[j], sh[j]), relu_lo);
    }
    const unsigned okm = ((pok >> P) & 1u) ? 0xffffffffu : 0u;
    uint4 pk;
    pk.x = wpack2(v[0], v[1]) & okm; pk.y = wpack2(v[2], v[3]) & okm;
    pk.z = wpack2(v[4], v[5]) & okm; pk.w = wpack2(v[6], v[7]) & okm;
    if (gtail) {
      pk.x &= pair_mask(gcb, a.Cg); pk.y &= pair_mask(gcb + 2, a.Cg); pk.z &= pair_mask(gcb + 4, a.Cg); pk.w &= pair_mask(gcb + 6, a.Cg);
    }
    *reinterpret_cast<uint4*>(gst + P * (G::RPP * G::BX * 64)) = pk;
  };
  auto issue = [&](int tile) {
    pn = tile / tpn;
    int t = tile % tpn;
    const int txi = t % a.tx; t /= a.tx;
    const int tyi = t % a.ty;
    const int tzi = t / a.ty;
    poz0 = tzi * TZ; poy0 = tyi * TY; pox0 = txi * 8;
    const float* dbase = reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.dn) + (long long)pn * a.dsn * (DBF ? 2 : 4));
    const int oy = poy0 + dyl, ox = pox0 + dxl;
    const bool yxok = oy < a.Hd && ox < a.Wd;
    const unsigned yxoff = __umul24((unsigned)min(oy, a.Hd - 1), dsh) + __umul24((unsigned)min(ox, a.Wd - 1), dsw);
    dok = 0u;
#pragma unroll
    for (int p = 0; p < DP; ++p) {
      const int oz = poz0 + dzl + p * (64 / (TY * 8));
      dok |= ((yxok && oz < a.Dd) ? 1u : 0u) << p;
      const unsigned off = __umul24((unsigned)min(oz, a.Dd - 1), dsd) + yxoff;
      dq[p] = oct8_ld<DBF>(dbase, off + dc_lo, off + dc_hi);
      if constexpr (NB == 2) dq[DP + p] = oct8_ld<DBF>(dbase, off + dc_lo1, off + dc_hi1);
    }
    pok = 0u;
    tile_x(pox0 * SI - 1);
    const float* gbase = reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.g) + (long long)pn * a.gsn * (GBF ? 2 : 4));
    static_for<0, PG>([&](auto pc) { load_g(pc, gbase, poz0 * SI - 1, poy0 * SI - 1); });
  };

  if (t0 < t1) issue(t0);
  for (int tile = t0; tile < t1; ++tile) {
    const int n = pn;
    // what the prefetch had no registers for comes in rounds of RB passes (all loads of a round in flight together);
    // the first round is requested before the prefetched part is committed
    constexpr int RB = GBF ? 12 : 4;
    constexpr int R1 = PG + RB < GP ? PG + RB : GP;
    const float* gbase = reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.g) + (long long)n * a.gsn * (GBF ? 2 : 4));
    const int iz0 = poz0 * SI - 1, iy0 = poy0 * SI - 1;
    static_for<PG, R1>([&](auto pc) { load_g(pc, gbase, iz0, iy0); });
    if (n != cn) {       // per-(n, channel) transform, once per batch item; kept in LDS (16 registers across the MFMA loop
                         // cost more than four LDS reads per tile); every thread of a chunk writes the same values
      if constexpr (NB == 1) {
        float s8[8], h8[8];
        if constexpr (TD) nl_coeff_vec<8>(a.td, n, a.Cd, dcb, s8, h8);
        else nl_coeff_vec<8>(a.tg, n, a.Cg, gcb, s8, h8);
        const int k8 = TD ? d8 : c8;
#pragma unroll
        for (int j = 0; j < 8; ++j) { coefl[8 * k8 + j] = s8[j]; coefl[32 + 8 * k8 + j] = h8[j]; }
      } else {
#pragma unroll
      for (int nb = 0; nb < (TD ? NB : 1); ++nb) {
        float s8[8], h8[8];
        if constexpr (TD) nl_coeff_vec<8>(a.td, n, a.Cd, dcb + 32 * nb, s8, h8);
        else nl_coeff_vec<8>(a.tg, n, a.Cg, gcb, s8, h8);
        const int k8 = TD ? d8 : c8;
#pragma unroll
        for (int j = 0; j < 8; ++j) { coefl[64 * nb + 8 * k8 + j] = s8[j]; coefl[64 * nb + 32 + 8 * k8 + j] = h8[j]; }
      }
      }
      cn = n;
      __syncthreads();
    }
    float sc[8], sh[8];
    // (the one-block form is kept statement for statement: the register allocation of the two-workgroup kernels is at its
    // limit, and the paired form's loop nest around the same statements cost them 100 bytes more spills, 206 -> 295 us)
    if constexpr (NB == 1) {
    {
      const float4* cq = reinterpret_cast<const float4*>(coefl + 8 * (TD ? d8 : c8));
      const float4 s0 = cq[0], s1 = cq[1], h0 = cq[8], h1 = cq[9];
      sc[0] = s0.x; sc[1] = s0.y; sc[2] = s0.z; sc[3] = s0.w; sc[4] = s1.x; sc[5] = s1.y; sc[6] = s1.z; sc[7] = s1.w;
      sh[0] = h0.x; sh[1] = h0.y; sh[2] = h0.z; sh[3] = h0.w; sh[4] = h1.x; sh[5] = h1.y; sh[6] = h1.z; sh[7] = h1.w;
    }
#pragma unroll
    for (int p = 0; p < DP; ++p) {
      float v[8];
      oct8_f8(dq[p], v);
      if constexpr (TD) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaxf(fmaf(v[j], sc[j], sh[j]), relu_lo);
      }
      const unsigned okm = ((dok >> p) & 1u) ? 0xffffffffu : 0u;
      if (want_db) {                                                  // bias gradient: fp32 sums of what is staged
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const unsigned cm = (dcb + j < a.Cd) ? okm : 0u;
          dbs[0][j] += __uint_as_float(__float_as_uint(v[j]) & cm);
        }
      }
      uint4 pk;
      pk.x = wpack2(v[0], v[1]) & okm; pk.y = wpack2(v[2], v[3]) & okm;
      pk.z = wpack2(v[4], v[5]) & okm; pk.w = wpack2(v[6], v[7]) & okm;
      if (dtail) {
        pk.x &= pair_mask(dcb, a.Cd); pk.y &= pair_mask(dcb + 2, a.Cd); pk.z &= pair_mask(dcb + 4, a.Cd); pk.w &= pair_mask(dcb + 6, a.Cd);
      }
      *reinterpret_cast<uint4*>(dst + p * 4096) = pk;
    }
    } else {
#define MMTTA_COEF_REGS(nb)                                                                                            \
    {                                                                                                                  \
      const float4* cq = reinterpret_cast<const float4*>(coefl + 64 * (nb) + 8 * (TD ? d8 : c8));                      \
      const float4 s0 = cq[0], s1 = cq[1], h0 = cq[8], h1 = cq[9];                                                     \
      sc[0] = s0.x; sc[1] = s0.y; sc[2] = s0.z; sc[3] = s0.w; sc[4] = s1.x; sc[5] = s1.y; sc[6] = s1.z; sc[7] = s1.w; \
      sh[0] = h0.x; sh[1] = h0.y; sh[2] = h0.z; sh[3] = h0.w; sh[4] = h1.x; sh[5] = h1.y; sh[6] = h1.z; sh[7] = h1.w; \
    }
    if constexpr (!TD || NB == 1) MMTTA_COEF_REGS(0)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
    if constexpr (TD && NB > 1) MMTTA_COEF_REGS(nb)
    const int dcn = dcb + 32 * nb;
#pragma unroll
    for (int p = 0; p < DP; ++p) {
      float v[8];
      oct8_f8(dq[nb * DP + p], v);
      if constexpr (TD) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaxf(fmaf(v[j], sc[j], sh[j]), relu_lo);
      }
      const unsigned okm = ((dok >> p) & 1u) ? 0xffffffffu : 0u;
      if (want_db) {                                                  // bias gradient: fp32 sums of what is staged
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const unsigned cm = (dcn + j < a.Cd) ? okm : 0u;
          dbs[nb][j] += __uint_as_float(__float_as_uint(v[j]) & cm);
        }
      }
      uint4 pk;
      pk.x = wpack2(v[0], v[1]) & okm; pk.y = wpack2(v[2], v[3]) & okm;
      pk.z = wpack2(v[4], v[5]) & okm; pk.w = wpack2(v[6], v[7]) & okm;
      if (dtail) {
        pk.x &= pair_mask(dcn, a.Cd); pk.y &= pair_mask(dcn + 2, a.Cd); pk.z &= pair_mask(dcn + 4, a.Cd); pk.w &= pair_mask(dcn + 6, a.Cd);
      }
      *reinterpret_cast<uint4*>(dst + nb * G::D_BYTES + p * 4096) = pk;
    }
    }
    }
    static_for<0, R1>([&](auto pc) { commit_g(pc, sc, sh); });
    if constexpr (R1 < GP) {
      constexpr int R2 = R1 + RB < GP ? R1 + RB : GP;
      static_for<R1, R2>([&](auto pc) { load_g(pc, gbase, iz0, iy0); });
      static_for<R1, R2>([&](auto pc) { commit_g(pc, sc, sh); });
      if constexpr (R2 < GP) {
        static_for<R2, GP>([&](auto pc) { load_g(pc, gbase, iz0, iy0); });
        static_for<R2, GP>([&](auto pc) { commit_g(pc, sc, sh); });
      }
    }
    __syncthreads();
    if (tile + 1 < t1) issue(tile + 1);                               // lands during the MFMAs below
    // ---- MFMAs.  Fragment f = (k step, tap slot) is read LA MFMAs before it multiplies (a ring of LA + 1 register sets;
    // a fence per MFMA, or the scheduler sinks every read to one MFMA ahead and the MFMA waits out the LDS latency)
    constexpr int LA = 3, NF = G::NK * 7;
    wbf16x8 ra[LA + 1], fb[2][NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) fb[0][nb] = MMTTA_TR_FRAG(dread, nb * G::D_BYTES);
#pragma unroll
    for (int f = 0; f < LA; ++f) ra[f] = MMTTA_TR_FRAG(gread[f % 7], G::row_off(f / 7));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const int ks = f / 7, j = f % 7;
      if (f + LA < NF) ra[(f + LA) % (LA + 1)] = MMTTA_TR_FRAG(gread[(f + LA) % 7], G::row_off((f + LA) / 7));
      if (j == 2 && ks + 1 < G::NK) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) fb[(ks + 1) & 1][nb] = MMTTA_TR_FRAG(dread, nb * G::D_BYTES + (ks + 1) * 1024);
      }
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
        acc[nb][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ra[f % (LA + 1)], fb[ks & 1][nb], acc[nb][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }
  const int sl = sx;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    const int tap = wave + 4 * j;
    if (tap < 27) {
      float* sb = a.slab + (((long long)sl * 27 + tap) * a.CGp + cg0) * a.CDp + cd0 + 32 * nb + r;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        sb[(long long)row * a.CDp] = acc[nb][j][i];
      }
    }
  }
  if (want_db) {                                    // thread (voxel slot, chunk d8) holds channels 8 d8 .. 8 d8 + 7
    float* red8 = lds;                              // the images are dead: the loop ended with a barrier
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      if (nb) __syncthreads();
#pragma unroll
      for (int c = 0; c < 8; ++c) red8[tid * 8 + c] = dbs[nb][c];
      __syncthreads();
      if (tid < 32) {
        float sacc = 0.f;
#pragma unroll 8
        for (int q = 0; q < 64; ++q) sacc += red8[((q << 2) | (tid >> 3)) * 8 + (tid & 7)];
        a.dbpart[(long long)sl * a.CDp + cd0 + 32 * nb + tid] = sacc;
      }
    }
  }
}

template <int TZ, int TY, int SI, bool GBF, bool DBF, bool TD, int NB = 1>
static int launch_wgrad_tr_t(const WArgs& a, int S, hipStream_t s) {
  using G = WTGeo<TZ, TY, SI>;
  auto kern = wgrad_tr_kernel<TZ, TY, SI, GBF, DBF, TD, NB>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  MMTTA_CHECK((a.CDp / 32) % NB == 0, MMTTA_ERR_INVALID, "wgrad: %d dense column blocks in groups of %d", a.CDp / 32, NB);
  dim3 grid(S, a.CGp / 32, a.CDp / 32 / NB);
  hipLaunchKernelGGL(kern, grid, dim3(256), G::lds_bytes(NB), s, a);
  return launch_status("conv wgrad bf16 (transposed reads)");
}

// ------------------------------------------------------------------ 1x1x1 weight gradient on transposed reads (bf16 operands)
// dw[cg][cd] = sum over voxels of x[v][cg] dy[v][cd]: no taps, no halo - a streaming kernel.  The fp32 form above walks
// 128-voxel tiles through load -> barrier -> 64 fp32 MFMAs -> barrier with a handful of workgroups (measured on the
// deep-fusion decoder: 33 -> 32 at 128^3 = 1.2 ms for 545 MB, 3.6 TFLOP/s).  Here both operands of a 4 x 8 x 8 tile sit in
// LDS as they sit in HBM ([voxel][32 channels] bf16), ds_read_b64_tr_b16 feeds v_mfma_f32_32x32x16_bf16, the four waves
// split the 16 k steps of a tile (one accumulator block each: registers are free for a full tile of prefetch) and write
// one slab each, as the fp32 kernel does; the loads of tile i+1 are requested before tile i's barrier.
template <bool GBF, bool DBF>
__global__ __launch_bounds__(256, 2) void wgrad_tr1_kernel(WArgs a) {
  constexpr int TZ = 4, TY = 8, NV = TZ * TY * 8, NP = NV * 4 / 256;      // 256 voxels, 4 items per thread and operand
  extern __shared__ float lds[];
  unsigned char* gl = reinterpret_cast<unsigned char*>(lds);
  unsigned char* dl = gl + NV * 64;
  float* coefl = reinterpret_cast<float*>(dl + NV * 64);                // [2][32] scale, shift of the module-input channels
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, r = lane & 31;
  const int cg0 = blockIdx.y * 32, cd0 = blockIdx.z * 32;
  const int lq = (lane & 15) >> 2, lp = lane & 3, lc = (lane >> 4) & 1;
  const unsigned char* gread = gl + (h * 8 + lq) * 64 + lc * 32 + lp * 8;
  const unsigned char* dread = dl + (h * 8 + lq) * 64 + lc * 32 + lp * 8;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float dbs[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) dbs[c] = 0.f;
  const bool want_db = a.dbpart != nullptr && blockIdx.y == 0;
  const bool td = a.convt != 0;                                         // the dense operand carries the norm-on-load

  const int vox0 = tid >> 2, c8 = tid & 3;                              // item p: voxel vox0 + 64 p (z = p), chunk c8
  unsigned char* gst = gl + vox0 * 64 + c8 * 16;
  unsigned char* dst = dl + vox0 * 64 + c8 * 16;
  const int yl = vox0 >> 3, xl = vox0 & 7;
  const int gcb = cg0 + 8 * c8, dcb = cd0 + 8 * c8;
  const unsigned gc_lo = GBF ? min(gcb, (a.Cg - 1) & ~7) : min(gcb, (a.Cg - 1) & ~3), gc_hi = min(gcb + 4, (a.Cg - 1) & ~3);
  const unsigned dc_lo = DBF ? min(dcb, (a.Cd - 1) & ~7) : min(dcb, (a.Cd - 1) & ~3), dc_hi = min(dcb + 4, (a.Cd - 1) & ~3);
  const bool gtail = (a.Cg & 7) != 0, dtail = (a.Cd & 7) != 0;
  auto pair_mask = [](int c, int C) { return (c < C ? 0xffffu : 0u) | (c + 1 < C ? 0xffff0000u : 0u); };
  const unsigned gsd = (unsigned)a.gsd, gsh = (unsigned)a.gsh, gsw = (unsigned)a.gsw;
  const unsigned dsd = (unsigned)a.dsd, dsh = (unsigned)a.dsh, dsw = (unsigned)a.dsw;

  // slab sx = slab (sx % S) of parameter set (sx / S): a set's slabs cover ITS batch items only, in the order a launch of
  // those items alone would use (the reduce kernels then sum a set's rows in that same order)
  const int sx = (int)xcd_contiguous_id(blockIdx.x, gridDim.x);
  const int qset = sx / a.S, sls = sx - qset * a.S;
  const int t0 = qset * a.tiles_set + sls * a.tiles_per_split;
  const int t1 = min((qset + 1) * a.tiles_set, t0 + a.tiles_per_split);
  const int tpn = a.tz * a.ty * a.tx;
  Oct8<GBF> gv[NP];
  Oct8<DBF> dq[NP];
  unsigned okbits = 0u;
  int pn = -1, cn = -1;
  auto issue = [&](int tile) {
    pn = tile / tpn;
    int t = tile % tpn;
    const int txi = t % a.tx; t /= a.tx;
    const int tyi = t % a.ty;
    const int tzi = t / a.ty;
    const int oz0 = tzi * TZ, oy = tyi * TY + yl, ox = txi * 8 + xl;
    const float* gbase = reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.g) + (long long)pn * a.gsn * (GBF ? 2 : 4));
    const float* dbase = reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.dn) + (long long)pn * a.dsn * (DBF ? 2 : 4));
    const bool yxok = oy < a.Hd && ox < a.Wd;
    const unsigned gyx = __umul24((unsigned)min(oy, a.Hd - 1), gsh) + __umul24((unsigned)min(ox, a.Wd - 1), gsw);
    const unsigned dyx = __umul24((unsigned)min(oy, a.Hd - 1), dsh) + __umul24((unsigned)min(ox, a.Wd - 1), dsw);
    okbits = 0u;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int oz = oz0 + p;
      okbits |= ((yxok && oz < a.Dd) ? 1u : 0u) << p;
      const unsigned go = __umul24((unsigned)min(oz, a.Dd - 1), gsd) + gyx, dof = __umul24((unsigned)min(oz, a.Dd - 1), dsd) + dyx;
      gv[p] = oct8_ld<GBF>(gbase, go + gc_lo, go + gc_hi);
      dq[p] = oct8_ld<DBF>(dbase, dof + dc_lo, dof + dc_hi);
    }
  };
  if (t0 < t1) issue(t0);
  for (int tile = t0; tile < t1; ++tile) {
    const int n = pn;
    if (n != cn) {
      float s8[8], h8[8];
      if (td) nl_coeff_vec<8>(a.td, n, a.Cd, dcb, s8, h8);
      else nl_coeff_vec<8>(a.tg, n, a.Cg, gcb, s8, h8);
#pragma unroll
      for (int j = 0; j < 8; ++j) { coefl[8 * c8 + j] = s8[j]; coefl[32 + 8 * c8 + j] = h8[j]; }
      cn = n;
      __syncthreads();
    }
    float sc[8], sh[8];
    {
      const float4* cq = reinterpret_cast<const float4*>(coefl + 8 * c8);
      const float4 s0 = cq[0], s1 = cq[1], h0 = cq[8], h1 = cq[9];
      sc[0] = s0.x; sc[1] = s0.y; sc[2] = s0.z; sc[3] = s0.w; sc[4] = s1.x; sc[5] = s1.y; sc[6] = s1.z; sc[7] = s1.w;
      sh[0] = h0.x; sh[1] = h0.y; sh[2] = h0.z; sh[3] = h0.w; sh[4] = h1.x; sh[5] = h1.y; sh[6] = h1.z; sh[7] = h1.w;
    }
    const float relu_lo = (td ? a.td.relu : a.tg.relu) ? 0.f : -__builtin_inff();
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const unsigned okm = ((okbits >> p) & 1u) ? 0xffffffffu : 0u;
      float v[8], u[8];
      oct8_f8(gv[p], v);
      oct8_f8(dq[p], u);
      if (td) {
#pragma unroll
        for (int j = 0; j < 8; ++j) u[j] = fmaxf(fmaf(u[j], sc[j], sh[j]), relu_lo);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaxf(fmaf(v[j], sc[j], sh[j]), relu_lo);
      }
      if (want_db) {
#pragma unroll
        for (int j = 0; j < 8; ++j) dbs[j] += __uint_as_float(__float_as_uint(u[j]) & ((dcb + j < a.Cd) ? okm : 0u));
      }
      uint4 pk, pd;
      pk.x = wpack2(v[0], v[1]) & okm; pk.y = wpack2(v[2], v[3]) & okm; pk.z = wpack2(v[4], v[5]) & okm; pk.w = wpack2(v[6], v[7]) & okm;
      pd.x = wpack2(u[0], u[1]) & okm; pd.y = wpack2(u[2], u[3]) & okm; pd.z = wpack2(u[4], u[5]) & okm; pd.w = wpack2(u[6], u[7]) & okm;
      if (gtail) { pk.x &= pair_mask(gcb, a.Cg); pk.y &= pair_mask(gcb + 2, a.Cg); pk.z &= pair_mask(gcb + 4, a.Cg); pk.w &= pair_mask(gcb + 6, a.Cg); }
      if (dtail) { pd.x &= pair_mask(dcb, a.Cd); pd.y &= pair_mask(dcb + 2, a.Cd); pd.z &= pair_mask(dcb + 4, a.Cd); pd.w &= pair_mask(dcb + 6, a.Cd); }
      *reinterpret_cast<uint4*>(gst + p * 4096) = pk;
      *reinterpret_cast<uint4*>(dst + p * 4096) = pd;
    }
    __syncthreads();
    if (tile + 1 < t1) issue(tile + 1);                               // lands during the reads / MFMAs below and the next commit
    wbf16x8 fa[4], fb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      fa[i] = MMTTA_TR_FRAG(gread, (wave + 4 * i) * 1024);
      fb[i] = MMTTA_TR_FRAG(dread, (wave + 4 * i) * 1024);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[i], acc, 0, 0, 0);
    __syncthreads();
  }
  const int sl = sx * 4 + wave;                                       // one slab per wave (the reduce kernels sum them)
  {
    float* sb = a.slab + ((long long)sl * a.CGp + cg0) * a.CDp + cd0 + r;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
      sb[(long long)row * a.CDp] = acc[i];
    }
  }
  if (want_db) {                                    // the workgroup's bias sums go to wave 0's slab row, zeros to the others
    float* red8 = lds;
#pragma unroll
    for (int c = 0; c < 8; ++c) red8[tid * 8 + c] = dbs[c];
    __syncthreads();
    if (tid < 32) {
      float sacc = 0.f;
#pragma unroll 8
      for (int q = 0; q < 64; ++q) sacc += red8[((q << 2) | (tid >> 3)) * 8 + (tid & 7)];
      a.dbpart[(long long)(sx * 4) * a.CDp + cd0 + tid] = sacc;
#pragma unroll
      for (int w2 = 1; w2 < 4; ++w2) a.dbpart[(long long)(sx * 4 + w2) * a.CDp + cd0 + tid] = 0.f;
    }
  }
}

template <bool GBF, bool DBF>
static int launch_wgrad_tr1_t(const WArgs& a, int S, hipStream_t s) {
  auto kern = wgrad_tr1_kernel<GBF, DBF>;
  dim3 grid(S, a.CGp / 32, a.CDp / 32);
  hipLaunchKernelGGL(kern, grid, dim3(256), 2 * 256 * 64 + 256, s, a);
  return launch_status("conv wgrad 1x1x1 bf16 (transposed reads)");
}

static int launch_wgrad_tr1(const WArgs& a, int S, hipStream_t s) {
  if (a.g_bf && a.d_bf) return launch_wgrad_tr1_t<true, true>(a, S, s);      // bf16-stored gradients (method.grad_storage)
  if (a.g_bf) return launch_wgrad_tr1_t<true, false>(a, S, s);
  if (a.d_bf) return launch_wgrad_tr1_t<false, true>(a, S, s);
  return launch_wgrad_tr1_t<false, false>(a, S, s);
}

// the module input (norm-on-load, possibly bf16-stored) is the gathered operand of a convolution and the dense one of a
// transposed convolution (stride 2 only); the other operand is a gradient: read as is, fp32- or (method.grad_storage: bf16,
// then next to a bf16-stored module input only) bf16-stored
template <int TZ, int TY, int SI>
static int launch_wgrad_tr(const WArgs& a, int S, hipStream_t s, int ncb = 1) {
  MMTTA_CHECK(a.gvec4 && a.dvec4, MMTTA_ERR_INVALID, "wgrad: transposed-read kernel selected for unaligned tensors");
  MMTTA_CHECK(ncb == 1 || (a.g_bf && a.d_bf), MMTTA_ERR_INVALID, "wgrad: paired column blocks exist for the all-bf16 forms");
  if (ncb == 2) {
    if constexpr (SI == 2) {
      if (a.convt) return launch_wgrad_tr_t<TZ, TY, SI, true, true, true, 2>(a, S, s);
    }
    MMTTA_CHECK(!a.convt, MMTTA_ERR_UNSUPPORTED, "wgrad: conv_transpose is stride 2 only");
    return launch_wgrad_tr_t<TZ, TY, SI, true, true, false, 2>(a, S, s);
  }
  if (a.convt) {
    MMTTA_CHECK(!a.g_bf || a.d_bf, MMTTA_ERR_UNSUPPORTED, "wgrad: a bf16-stored gradient needs a bf16-stored module input");
    if constexpr (SI == 2) {
      if (a.g_bf) return launch_wgrad_tr_t<TZ, TY, SI, true, true, true>(a, S, s);
      if (a.d_bf) return launch_wgrad_tr_t<TZ, TY, SI, false, true, true>(a, S, s);
      return launch_wgrad_tr_t<TZ, TY, SI, false, false, true>(a, S, s);
    }
    return MMTTA_ERR_UNSUPPORTED;
  }
  MMTTA_CHECK(!a.d_bf || a.g_bf, MMTTA_ERR_UNSUPPORTED, "wgrad: a bf16-stored gradient needs a bf16-stored module input");
  if (a.d_bf) return launch_wgrad_tr_t<TZ, TY, SI, true, true, false>(a, S, s);
  if (a.g_bf) return launch_wgrad_tr_t<TZ, TY, SI, true, false, false>(a, S, s);
  return launch_wgrad_tr_t<TZ, TY, SI, false, false, false>(a, S, s);
}

// ------------------------------------------------------------------ small-channel weight gradient
// When one side of the layer has <= 4 channels (first layers: Cin = 1..4; last layers: Cout = 1..3) the
// 32x32 (cg x cd) blocking above would pad it 8-32x.  Here the small tensor Q is the GATHERED one and its
// (tap, channel) pairs are flattened into the MFMA row index (27*4 = 108 <= 128 rows = 4 waves x 32), the
// big tensor P is dense:      S[row=(tap,cs)][col=cb] = sum_o T(Q[o*si + tap - 1][cs]) * T(P[o][cb])
// One MFMA per voxel pair and wave instead of seven, and no padded channels.
struct W2Args {
  const float* q; long long qsn, qsd, qsh, qsw; int Cs, Dq, Hq, Wq; NL tq;
  const float* p; long long psn, psd, psh, psw; int Cb, Dp, Hp, Wp; NL tp;
  int si, ntaps;
  float* slab;     // [nsl][128][CBp]
  float* dbpart;   // [nsl][CBp] or null: per-channel sums of P (bias gradient when cb is the output channel)
  int tz, ty, tx, tiles, tiles_per_split, CBp;
  int S, tiles_set;   // slabs and tiles per parameter set
  int qvec4, pvec4;
  int p_bf;        // the dense tensor P is bf16-stored (forward activation of bf16 precision)
  int p_thin;      // wgrad_thin_tr_kernel: P has <= 4 channels too (fp32 voxels of 16 bytes)
  int q_bf;        // wgrad_thin_tr_kernel: the thin gathered tensor Q is bf16-stored (8 bytes a voxel)
  int bf;          // bf16 precision mode: operands rounded to bf16, 16 voxels per v_mfma_f32_32x32x16_bf16
};

// SI (gather stride) and T27 (27 taps / 1 tap) are template parameters: the box extents divide every staged voxel's index,
// and a run-time divisor is a ~30-instruction division (36 of them per tile and thread before)
template <bool PBF, int SI, bool T27>
__global__ __launch_bounds__(256) void wgrad_small_kernel(W2Args a) {
  extern __shared__ float lds[];
  constexpr int TZ = 4, TY = 4, TX = 8, MT = 128;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, r = lane & 31;
  constexpr int NT = T27 ? 27 : 1;
  constexpr int ext = T27 ? 2 : 0;
  constexpr int BZ = (TZ - 1) * SI + ext + 1, BY = (TY - 1) * SI + ext + 1, BX = (TX - 1) * SI + ext + 1;
  constexpr int boxvox = BZ * BY * BX;
  constexpr int dmin = T27 ? -1 : 0;
  float* ql = lds;                  // [boxvox][4]
  float* pl = lds + boxvox * 4;     // [MT][32]
  const int cb0 = blockIdx.y * 32;
  const int nrows = NT * a.Cs;
  const int rowid = wave * 32 + r;
  int toffl = 0;
  if (rowid < nrows) {
    const int tap = rowid / a.Cs, cs = rowid % a.Cs;
    toffl = (!T27 ? 0 : (((tap / 9) * BY + ((tap / 3) % 3)) * BX + (tap % 3))) * 4 + cs;
  }
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float dbacc = 0.f;
  const int sx = (int)xcd_contiguous_id(blockIdx.x, gridDim.x);                       // one part of the volume per XCD
  const int qset = sx / a.S, sls = sx - qset * a.S;                                   // (parameter set, slab of the set)
  const int t0 = qset * a.tiles_set + sls * a.tiles_per_split;
  const int t1 = min((qset + 1) * a.tiles_set, t0 + a.tiles_per_split);
  const int tpn = a.tz * a.ty * a.tx;
  // Both operands 16-byte addressable (every layer of the shipped networks): the NEXT tile's global loads are issued
  // before this tile's MFMAs and land while they run (a workgroup walked load -> barrier -> MFMA -> barrier with one
  // workgroup per CU: every tile paid a full memory latency).  Otherwise the loads sit in the staging step itself.
  constexpr int NQ = (boxvox + 255) / 256;          // 6 for the stride-2 box (9 x 9 x 17)
  const bool piped = a.qvec4 && a.pvec4;
  float4 raw[NQ], praw[MT / 32];
  float qsc[4], qsh[4], psc[4], psh[4];
  const int pcv = tid & 7, pc = cb0 + pcv * 4;
  const int pcl4 = min(pc, (a.Cb - 1) & ~3);      // clamped channel group (16-byte aligned)
  int n_coef = -1;
  auto decode = [&](int tile, int& n, int& oz0, int& oy0, int& ox0) {
    n = tile / tpn;
    int t = tile % tpn;
    const int txi = t % a.tx; t /= a.tx;
    const int tyi = t % a.ty;
    const int tzi = t / a.ty;
    oz0 = tzi * TZ; oy0 = tyi * TY; ox0 = txi * TX;
  };
  auto issue = [&](int tile) {
    int n, oz0, oy0, ox0;
    decode(tile, n, oz0, oy0, ox0);
    const int iz0 = oz0 * SI + dmin, iy0 = oy0 * SI + dmin, ix0 = ox0 * SI + dmin;
    const float* qb = a.q + (long long)n * a.qsn;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int bv = min(tid + 256 * q, boxvox - 1);
      const int bx = bv % BX, by = (bv / BX) % BY, bz = bv / (BX * BY);
      const int iz = min(max(iz0 + bz, 0), a.Dq - 1), iy = min(max(iy0 + by, 0), a.Hq - 1), ix = min(max(ix0 + bx, 0), a.Wq - 1);
      raw[q] = *reinterpret_cast<const float4*>(qb + iz * a.qsd + iy * a.qsh + ix * a.qsw);
    }
    const long long pbo = (long long)n * a.psn;
#pragma unroll
    for (int q = 0; q < MT / 32; ++q) {
      const int v = (tid >> 3) + 32 * q;
      const int xl = v % TX, yl = (v / TX) % TY, zl = v / (TX * TY);
      const int oz = min(oz0 + zl, a.Dp - 1), oy = min(oy0 + yl, a.Hp - 1), ox = min(ox0 + xl, a.Wp - 1);
      praw[q] = ld4_t<PBF>(a.p, pbo + (long long)oz * a.psd + (long long)oy * a.psh + (long long)ox * a.psw + pcl4);
    }
    if (n != n_coef) {
      nl_coeff_vec<4>(a.tq, n, a.Cs, 0, qsc, qsh);
      nl_coeff_vec<4>(a.tp, n, a.Cb, pc, psc, psh);
      n_coef = n;
    }
  };
  auto commit = [&](int tile) {
    int n, oz0, oy0, ox0;
    decode(tile, n, oz0, oy0, ox0);
    const int iz0 = oz0 * SI + dmin, iy0 = oy0 * SI + dmin, ix0 = ox0 * SI + dmin;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int bv = tid + 256 * q;
      if (bv < boxvox) {
        const int bx = bv % BX, by = (bv / BX) % BY, bz = bv / (BX * BY);
        const int iz = iz0 + bz, iy = iy0 + by, ix = ix0 + bx;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((unsigned)iz < (unsigned)a.Dq && (unsigned)iy < (unsigned)a.Hq && (unsigned)ix < (unsigned)a.Wq) {
          v.x = nl_apply(raw[q].x, qsc[0], qsh[0], a.tq.relu);
          v.y = a.Cs > 1 ? nl_apply(raw[q].y, qsc[1], qsh[1], a.tq.relu) : 0.f;
          v.z = a.Cs > 2 ? nl_apply(raw[q].z, qsc[2], qsh[2], a.tq.relu) : 0.f;
          v.w = a.Cs > 3 ? nl_apply(raw[q].w, qsc[3], qsh[3], a.tq.relu) : 0.f;
        }
        *reinterpret_cast<float4*>(ql + bv * 4) = v;
      }
    }
#pragma unroll
    for (int q = 0; q < MT / 32; ++q) {
      const int v = (tid >> 3) + 32 * q;
      const int xl = v % TX, yl = (v / TX) % TY, zl = v / (TX * TY);
      const int oz = oz0 + zl, oy = oy0 + yl, ox = ox0 + xl;
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
      if (oz < a.Dp && oy < a.Hp && ox < a.Wp && pc < a.Cb) {
        const float4 x4 = praw[q];
        o.x = nl_apply(x4.x, psc[0], psh[0], a.tp.relu);
        o.y = (pc + 1 < a.Cb) ? nl_apply(x4.y, psc[1], psh[1], a.tp.relu) : 0.f;
        o.z = (pc + 2 < a.Cb) ? nl_apply(x4.z, psc[2], psh[2], a.tp.relu) : 0.f;
        o.w = (pc + 3 < a.Cb) ? nl_apply(x4.w, psc[3], psh[3], a.tp.relu) : 0.f;
      }
      *reinterpret_cast<float4*>(pl + v * 32 + pcv * 4) = o;
    }
  };
  if (piped && t0 < t1) issue(t0);
  for (int tile = t0; tile < t1; ++tile) {
    if (piped) {
      commit(tile);
    } else {
    const int n = tile / tpn;
    int t = tile % tpn;
    const int txi = t % a.tx; t /= a.tx;
    const int tyi = t % a.ty;
    const int tzi = t / a.ty;
    const int oz0 = tzi * TZ, oy0 = tyi * TY, ox0 = txi * TX;
    const int iz0 = oz0 * SI + dmin, iy0 = oy0 * SI + dmin, ix0 = ox0 * SI + dmin;
    {  // Q box: up to 4 channels per voxel
      const float* qb = a.q + (long long)n * a.qsn;
      float sc[4], sh[4];
      nl_coeff_vec<4>(a.tq, n, a.Cs, 0, sc, sh);
      // all of the thread's box voxels are loaded (from clamped addresses) before the first is used
      if (a.qvec4) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int bv = min(tid + 256 * q, boxvox - 1);
          const int bx = bv % BX, by = (bv / BX) % BY, bz = bv / (BX * BY);
          const int iz = min(max(iz0 + bz, 0), a.Dq - 1), iy = min(max(iy0 + by, 0), a.Hq - 1), ix = min(max(ix0 + bx, 0), a.Wq - 1);
          raw[q] = *reinterpret_cast<const float4*>(qb + iz * a.qsd + iy * a.qsh + ix * a.qsw);
        }
      } else {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int bv = min(tid + 256 * q, boxvox - 1);
          const int bx = bv % BX, by = (bv / BX) % BY, bz = bv / (BX * BY);
          const int iz = min(max(iz0 + bz, 0), a.Dq - 1), iy = min(max(iy0 + by, 0), a.Hq - 1), ix = min(max(ix0 + bx, 0), a.Wq - 1);
          const float* src = qb + iz * a.qsd + iy * a.qsh + ix * a.qsw;
          raw[q].x = src[0];
          raw[q].y = src[min(1, a.Cs - 1)]; raw[q].z = src[min(2, a.Cs - 1)]; raw[q].w = src[min(3, a.Cs - 1)];
        }
      }
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int bv = tid + 256 * q;
        if (bv < boxvox) {
          const int bx = bv % BX, by = (bv / BX) % BY, bz = bv / (BX * BY);
          const int iz = iz0 + bz, iy = iy0 + by, ix = ix0 + bx;
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if ((unsigned)iz < (unsigned)a.Dq && (unsigned)iy < (unsigned)a.Hq && (unsigned)ix < (unsigned)a.Wq) {
            v.x = nl_apply(raw[q].x, sc[0], sh[0], a.tq.relu);
            v.y = a.Cs > 1 ? nl_apply(raw[q].y, sc[1], sh[1], a.tq.relu) : 0.f;
            v.z = a.Cs > 2 ? nl_apply(raw[q].z, sc[2], sh[2], a.tq.relu) : 0.f;
            v.w = a.Cs > 3 ? nl_apply(raw[q].w, sc[3], sh[3], a.tq.relu) : 0.f;
          }
          *reinterpret_cast<float4*>(ql + bv * 4) = v;
        }
      }
    }
    {  // P tile
      const float* pb = a.p;
      const long long pbo = (long long)n * a.psn;
      if (a.pvec4) {
        const int cv = tid & 7, c = cb0 + cv * 4;
        float sc[4], sh[4];
        nl_coeff_vec<4>(a.tp, n, a.Cb, c, sc, sh);
        const int cl4 = min(c, (a.Cb - 1) & ~3);           // clamped channel group (16-byte aligned)
#pragma unroll
        for (int q = 0; q < MT / 32; ++q) {
          const int v = (tid >> 3) + 32 * q;
          const int xl = v % TX, yl = (v / TX) % TY, zl = v / (TX * TY);
          const int oz = min(oz0 + zl, a.Dp - 1), oy = min(oy0 + yl, a.Hp - 1), ox = min(ox0 + xl, a.Wp - 1);
          praw[q] = ld4_t<PBF>(pb, pbo + (long long)oz * a.psd + (long long)oy * a.psh + (long long)ox * a.psw + cl4);
        }
#pragma unroll
        for (int q = 0; q < MT / 32; ++q) {
          const int v = (tid >> 3) + 32 * q;
          const int xl = v % TX, yl = (v / TX) % TY, zl = v / (TX * TY);
          const int oz = oz0 + zl, oy = oy0 + yl, ox = ox0 + xl;
          float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
          if (oz < a.Dp && oy < a.Hp && ox < a.Wp && c < a.Cb) {
            const float4 x4 = praw[q];
            o.x = nl_apply(x4.x, sc[0], sh[0], a.tp.relu);
            o.y = (c + 1 < a.Cb) ? nl_apply(x4.y, sc[1], sh[1], a.tp.relu) : 0.f;
            o.z = (c + 2 < a.Cb) ? nl_apply(x4.z, sc[2], sh[2], a.tp.relu) : 0.f;
            o.w = (c + 3 < a.Cb) ? nl_apply(x4.w, sc[3], sh[3], a.tp.relu) : 0.f;
          }
          *reinterpret_cast<float4*>(pl + v * 32 + cv * 4) = o;
        }
      } else {
        const int cc = tid & 31, c = cb0 + cc;
        float sc = 0.f, sh = 0.f;
        nl_coeff_vec<1>(a.tp, n, a.Cb, c, &sc, &sh);
        for (int v = tid >> 5; v < MT; v += 8) {
          const int xl = v % TX, yl = (v / TX) % TY, zl = v / (TX * TY);
          const int oz = oz0 + zl, oy = oy0 + yl, ox = ox0 + xl;
          float val = 0.f;
          if (oz < a.Dp && oy < a.Hp && ox < a.Wp && c < a.Cb)
            val = nl_apply(ld1_t<PBF>(pb, pbo + (long long)oz * a.psd + (long long)oy * a.psh + (long long)ox * a.psw + c), sc, sh, a.tp.relu);
          pl[v * 32 + cc] = val;
        }
      }
    }
    }
    __syncthreads();
    if (piped && tile + 1 < t1) issue(tile + 1);
    // voxel v = 2*kk + h: the pair shares (zl, yl) and differs by one x step, so the box offset is a wave-uniform
    // (scalar) term per kk plus a lane term that does not change inside the tile loop
    if (a.bf) {
      // k = h*8 + e of MFMA step kk is voxel 16*kk + 8*h + e: with TX = 8 that is the x-row e = 0..7 at
      // (yl, zl) = ((2*kk + h) % TY, (2*kk + h) / TY): 8 gathered values per operand, packed to bf16
      const int xs = SI * 4;
#pragma unroll 2
      for (int kk = 0; kk < MT / 16; ++kk) {
        const int rowi = 2 * kk + h;
        const int yl = rowi % TY, zl = rowi / TY;
        const float* qp = ql + (((zl * SI) * BY + yl * SI) * BX) * 4 + toffl;
        const float* pp = pl + (rowi * 8) * 32 + r;
        float av[8], bv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { av[e] = qp[e * xs]; bv[e] = pp[e * 32]; }
        wbf16x8 af, bf;
#pragma unroll
        for (int e = 0; e < 8; ++e) { af[e] = (__bf16)av[e]; bf[e] = (__bf16)bv[e]; dbacc += bv[e]; }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc, 0, 0, 0);
      }
    } else {
    const float* qlane = ql + toffl + h * SI * 4;
    const float* plane = pl + h * 32 + r;
#pragma unroll 8
    for (int kk = 0; kk < MT / 2; ++kk) {
      const int xl0 = (2 * kk) % TX, yl = ((2 * kk) / TX) % TY, zl = (2 * kk) / (TX * TY);
      const int gaddr = (((zl * SI) * BY + yl * SI) * BX + xl0 * SI) * 4;
      const float av = qlane[gaddr];
      const float b = plane[kk * 64];
      dbacc += b;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b, acc, 0, 0, 0);
    }
    }
    __syncthreads();
  }
  const int sl = sx;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int row = wave * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
    if (row < nrows) a.slab[((long long)sl * 128 + row) * a.CBp + cb0 + r] = acc[i];
  }
  if (a.dbpart != nullptr && wave == 0) {
    dbacc += __shfl_xor(dbacc, 32, 64);
    if (h == 0) a.dbpart[(long long)sl * a.CBp + cb0 + r] = dbacc;
  }
}


// ------------------------------------------------------------------ thin-layer weight gradient on transposed reads
// The same contraction as wgrad_small_kernel's bf16 branch - S[(tap, cs)][cb] = sum_o Q[o*si + tap - 1][cs] * P[o][cb], 27 taps,
// operands rounded to bf16, v_mfma_f32_32x32x16_bf16 - with BOTH operands staged once as bf16 rows and read through
// ds_read_b64_tr_b16.  wgrad_small_kernel kept fp32 images and built every fragment from 16 ds_read_b32 + 8 packs: 95
// vector instructions per MFMA (profiles/r03a_pmc_sq_per_kernel.txt), which is what its launches were bound by.  Here
//  * P sits in LDS as it sits in HBM ([voxel][32 channels] bf16, 64-byte rows, NCB column blocks side by side): its fragment
//    is wgrad_tr_kernel's - two transposed reads 4 voxels apart;
//  * Q's box is [voxel][4 channels] bf16, 8 bytes a voxel.  A transposed read takes 4 rows x 16 columns from 16 lane
//    addresses (row = (lane & 15) >> 2, 8-byte chunk = lane & 3) and the hardware does not care where a chunk lies: chunk
//    lp of lane half lc is pointed at TAP 4 lc + lp of the wave's 8 taps, so the 32 columns a wave's fragment holds are
//    (tap, channel) = (column >> 2, column & 3) - the gather costs no instruction.  Wave w owns taps 8 w .. 8 w + 7 (the 5
//    slots past tap 26 re-read tap 26 and are dropped), one accumulator per column block;
//  * a k step is 16 output voxels = two x-rows of the 4 x 4 x 8 tile, as in wgrad_small_kernel.
// Staging: a thread's box voxels / tile items are the same for every tile (decoded once); offsets are 32-bit elements
// (host check); the next tile's loads are issued before this tile's MFMAs.
// PM: storage of the dense operand P - 0: fp32, 8-channel items; 1: bf16, 8-channel items; 2: P is thin as well (<= 4
// channels, fp32 voxels of 16 bytes: the full-resolution R -> R convolutions, both sides <= 4 channels): its tile rows
// hold 4 live columns, the other 28 are zeroed once (the matrix cores are idle either way; what counts is that a voxel
// costs a load, two packs and a store instead of wgrad_tiny_kernel's 243 FMAs on the vector ALU).
// QBF: the thin gathered tensor is bf16-stored, 8 bytes a voxel (the network input of bf16 precision).
template <int SI, int PM, int NCB, bool QBF = false>
__global__ __launch_bounds__(256, 2) void wgrad_thin_tr_kernel(W2Args a) {
  extern __shared__ float lds[];
  constexpr bool PBF = PM == 1, PTHIN = PM >= 2, PTBF = PM == 3;      // PM 3: the thin P is bf16-stored (8-byte voxels)
  static_assert(!PTHIN || NCB == 1, "a thin P is one column block");
  // stride 1: 4 x 8 x 8 tiles (box 6 x 10 x 10 = 2.3 voxels read per output voxel, 2.8 with 4 x 4 x 8; twice the MFMAs
  // behind one round of loads); stride 2: 4 x 4 x 8 (the 9 x 9 x 17 box is 6 staging passes already)
  constexpr int TZ = 4, TY = SI == 1 ? 8 : 4, TX = 8, MT = TZ * TY * TX;
  constexpr int BZ = (TZ - 1) * SI + 3, BY = (TY - 1) * SI + 3, BX = (TX - 1) * SI + 3;
  constexpr int boxvox = BZ * BY * BX;
  constexpr int QBYTES = (boxvox * 8 + 63) / 64 * 64;
  constexpr int NQ = (boxvox + 255) / 256;
  constexpr int NI = PTHIN ? 1 : MT * 4 * NCB / 256;  // items of the dense tile per thread (PTHIN: threads 0..127, one voxel each)
  unsigned char* ql = reinterpret_cast<unsigned char*>(lds);
  unsigned char* pl = ql + QBYTES;                    // NCB planes of [128 voxels][32 channels] bf16
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, r = lane & 31;
  const int lq = (lane & 15) >> 2, lp = lane & 3, lc = (lane >> 4) & 1;
  const int cb0 = blockIdx.y * 32 * NCB;

  // operand addresses (lane parts)
  const int tapl = min(wave * 8 + lc * 4 + lp, 26);
  const unsigned char* aread = ql + ((((tapl / 9) + 0) * BY + (tapl / 3) % 3 + h * SI) * BX + tapl % 3 + lq * SI) * 8;
  const unsigned char* bread = pl + (h * 8 + lq) * 64 + lc * 32 + lp * 8;
  f32x16 acc[NCB];
#pragma unroll
  for (int c = 0; c < NCB; ++c)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;

  // staging roles (tile-independent)
  int qpos[NQ];                                        // box voxel of pass q: bz << 16 | by << 8 | bx
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int bv = min(tid + 256 * q, boxvox - 1);
    qpos[q] = ((bv / (BX * BY)) << 16) | (((bv / BX) % BY) << 8) | (bv % BX);
  }
  const int c8 = PTHIN ? 0 : tid % (4 * NCB);                   // item p: voxel pv0 + p * (256 / (4 NCB)), channel chunk c8
  const int pv0 = PTHIN ? (tid & (MT - 1)) : tid / (4 * NCB);
  const int pc = cb0 + 8 * c8;
  const bool pcok = PTHIN ? tid < MT : pc < a.Cb;               // (Cb % 8 == 0: a chunk is inside or outside as a whole)
  const unsigned pcl = PTHIN ? 0u : (unsigned)min(pc, a.Cb - 8);
  unsigned char* pst = pl + (c8 >> 2) * (MT * 64) + pv0 * 64 + (c8 & 3) * 16;
  if constexpr (PTHIN) {                                        // columns 4..31 of the tile rows: zero for the whole launch
    for (int i = tid; i < MT * 7; i += 256) *reinterpret_cast<uint2*>(pl + (i / 7) * 64 + 8 + (i % 7) * 8) = make_uint2(0u, 0u);
  }
  const unsigned qsd = (unsigned)a.qsd, qsh = (unsigned)a.qsh, qsw = (unsigned)a.qsw;
  const unsigned psd = (unsigned)a.psd, psh = (unsigned)a.psh, psw = (unsigned)a.psw;
  const float qlo = a.tq.relu ? 0.f : -__builtin_inff(), plo = a.tp.relu ? 0.f : -__builtin_inff();

  const int sx = (int)xcd_contiguous_id(blockIdx.x, gridDim.x);
  const int qset = sx / a.S, sls = sx - qset * a.S;
  const int t0 = qset * a.tiles_set + sls * a.tiles_per_split;
  const int t1 = min((qset + 1) * a.tiles_set, t0 + a.tiles_per_split);
  const int tpn = a.tz * a.ty * a.tx;
  float4 raw[NQ];
  Oct8<PBF> pit[NI];
  float4 pthin = make_float4(0.f, 0.f, 0.f, 0.f);
  unsigned qok = 0u, pok = 0u;
  float qsc[4], qsf[4], psc[8], psf[8], dbs[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) dbs[j] = 0.f;
  int n_coef = -1;
  const bool want_db = a.dbpart != nullptr;

  auto issue = [&](int tile) {
    const int n = tile / tpn;
    int t = tile % tpn;
    const int txi = t % a.tx; t /= a.tx;
    const int tyi = t % a.ty;
    const int tzi = t / a.ty;
    const int oz0 = tzi * TZ, oy0 = tyi * TY, ox0 = txi * TX;
    const int iz0 = oz0 * SI - 1, iy0 = oy0 * SI - 1, ix0 = ox0 * SI - 1;
    const float* qb = QBF ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(a.q) + (long long)n * a.qsn)
                          : a.q + (long long)n * a.qsn;
    qok = 0u;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int iz = iz0 + (qpos[q] >> 16), iy = iy0 + ((qpos[q] >> 8) & 255), ix = ix0 + (qpos[q] & 255);
      const bool ok = (unsigned)iz < (unsigned)a.Dq && (unsigned)iy < (unsigned)a.Hq && (unsigned)ix < (unsigned)a.Wq &&
                      tid + 256 * q < boxvox;
      qok |= (ok ? 1u : 0u) << q;
      const unsigned off = (unsigned)min(max(iz, 0), a.Dq - 1) * qsd + (unsigned)min(max(iy, 0), a.Hq - 1) * qsh +
                           (unsigned)min(max(ix, 0), a.Wq - 1) * qsw;
      if constexpr (QBF) {          // two dwords of the raw register carry the four bf16 channels until they are committed
        const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(qb) + off);
        raw[q].x = __uint_as_float(u.x); raw[q].y = __uint_as_float(u.y);
      } else {
        raw[q] = *reinterpret_cast<const float4*>(qb + off);
      }
    }
    const float* pb = reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.p) + (long long)n * a.psn * ((PBF || PTBF) ? 2 : 4));
    pok = 0u;
#pragma unroll
    for (int p = 0; p < NI; ++p) {
      const int v = pv0 + p * (256 / (4 * NCB));
      const int oz = oz0 + v / (TX * TY), oy = oy0 + (v / TX) % TY, ox = ox0 + v % TX;
      const bool ok = pcok && oz < a.Dp && oy < a.Hp && ox < a.Wp;
      pok |= (ok ? 1u : 0u) << p;
      const unsigned off = (unsigned)min(oz, a.Dp - 1) * psd + (unsigned)min(oy, a.Hp - 1) * psh + (unsigned)min(ox, a.Wp - 1) * psw + pcl;
      if constexpr (PTBF) {
        const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(pb) + off);
        pthin.x = __uint_as_float(u.x); pthin.y = __uint_as_float(u.y);
      } else if constexpr (PTHIN) {
        pthin = *reinterpret_cast<const float4*>(pb + off);
      } else {
        pit[p] = oct8_ld<PBF>(pb, off, off + 4);
      }
    }
    if (n != n_coef) {
      nl_coeff_vec<4>(a.tq, n, a.Cs, 0, qsc, qsf);
      nl_coeff_vec<8>(a.tp, n, a.Cb, (int)pcl, psc, psf);
      n_coef = n;
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      if (tid + 256 * q < boxvox) {
        const unsigned okm = ((qok >> q) & 1u) ? 0xffffffffu : 0u;
        if constexpr (QBF) {
          const unsigned ux = __float_as_uint(raw[q].x), uy = __float_as_uint(raw[q].y);
          raw[q] = make_float4(bf16_bits_to_f32(ux & 0xffffu), __uint_as_float(ux & 0xffff0000u),
                               bf16_bits_to_f32(uy & 0xffffu), __uint_as_float(uy & 0xffff0000u));
        }
        const float v0 = fmaxf(fmaf(raw[q].x, qsc[0], qsf[0]), qlo);
        const float v1 = a.Cs > 1 ? fmaxf(fmaf(raw[q].y, qsc[1], qsf[1]), qlo) : 0.f;
        const float v2 = a.Cs > 2 ? fmaxf(fmaf(raw[q].z, qsc[2], qsf[2]), qlo) : 0.f;
        const float v3 = a.Cs > 3 ? fmaxf(fmaf(raw[q].w, qsc[3], qsf[3]), qlo) : 0.f;
        uint2 pk;
        pk.x = wpack2(v0, v1) & okm; pk.y = wpack2(v2, v3) & okm;
        *reinterpret_cast<uint2*>(ql + (tid + 256 * q) * 8) = pk;
      }
    }
    if constexpr (PTHIN) {
      if (tid < MT) {
        const unsigned okm = (pok & 1u) ? 0xffffffffu : 0u;
        if constexpr (PTBF) {
          const unsigned ux = __float_as_uint(pthin.x), uy = __float_as_uint(pthin.y);
          pthin = make_float4(bf16_bits_to_f32(ux & 0xffffu), __uint_as_float(ux & 0xffff0000u),
                              bf16_bits_to_f32(uy & 0xffffu), __uint_as_float(uy & 0xffff0000u));
        }
        const float raw4[4] = {pthin.x, pthin.y, pthin.z, pthin.w};
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j] = __uint_as_float(__float_as_uint(fmaxf(fmaf(raw4[j], psc[j], psf[j]), plo)) & (j < a.Cb ? okm : 0u));
          dbs[j] += v[j];
        }
        *reinterpret_cast<uint2*>(pst) = make_uint2(wpack2(v[0], v[1]), wpack2(v[2], v[3]));
      }
    }
#pragma unroll
    for (int p = 0; p < (PTHIN ? 0 : NI); ++p) {
      float v[8];
      oct8_f8(pit[p], v);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = fmaxf(fmaf(v[j], psc[j], psf[j]), plo);
      const unsigned okm = ((pok >> p) & 1u) ? 0xffffffffu : 0u;
      if (want_db) {
#pragma unroll
        for (int j = 0; j < 8; ++j) dbs[j] += __uint_as_float(__float_as_uint(v[j]) & okm);
      }
      uint4 pk;
      pk.x = wpack2(v[0], v[1]) & okm; pk.y = wpack2(v[2], v[3]) & okm;
      pk.z = wpack2(v[4], v[5]) & okm; pk.w = wpack2(v[6], v[7]) & okm;
      *reinterpret_cast<uint4*>(pst + p * (256 / (4 * NCB)) * 64) = pk;
    }
  };

  if (t0 < t1) issue(t0);
  for (int tile = t0; tile < t1; ++tile) {
    commit();
    __syncthreads();
    if (tile + 1 < t1) issue(tile + 1);                 // lands during the MFMAs below
#pragma unroll
    for (int kk = 0; kk < MT / 16; ++kk) {
      const int roff = ((((2 * kk) / TY) * SI * BY + ((2 * kk) % TY) * SI) * BX) * 8;
      const wbf16x8 af = __builtin_bit_cast(wbf16x8, __builtin_shufflevector(
          __builtin_amdgcn_ds_read_tr16_b64_v4i16((wlds4_t)(aread + roff)),
          __builtin_amdgcn_ds_read_tr16_b64_v4i16((wlds4_t)(aread + roff + 4 * SI * 8)), 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
      for (int c = 0; c < NCB; ++c) {
        const wbf16x8 bf = MMTTA_TR_FRAG(bread, c * (MT * 64) + kk * 1024);
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[c], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  const int sl = sx;
#pragma unroll
  for (int c = 0; c < NCB; ++c)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row32 = (i & 3) + 8 * (i >> 2) + 4 * h;         // (tap of the wave, channel) = (row32 >> 2, row32 & 3)
      const int tap = wave * 8 + (row32 >> 2), cs = row32 & 3;
      if (tap < 27 && cs < a.Cs) a.slab[((long long)sl * 128 + tap * a.Cs + cs) * a.CBp + cb0 + c * 32 + r] = acc[c][i];
    }
  if (want_db) {                                    // thread (voxel slot, chunk c8) holds channels cb0 + 8 c8 .. + 7
    float* red8 = lds;                              // the images are dead: the loop ended with a barrier
#pragma unroll
    for (int j = 0; j < 8; ++j) red8[tid * 8 + j] = dbs[j];
    __syncthreads();
    if (tid < 32 * NCB) {
      float sacc = 0.f;
      if constexpr (PTHIN) {                        // threads 0..127 hold channels 0..3 of their voxel
        if (tid < 4)
          for (int m = 0; m < MT; ++m) sacc += red8[m * 8 + tid];
      } else {
        const int k8 = tid >> 3, j = tid & 7;
        for (int m = 0; m < 256 / (4 * NCB); ++m) sacc += red8[(k8 + 4 * NCB * m) * 8 + j];
      }
      a.dbpart[(long long)sl * a.CBp + cb0 + tid] = sacc;
    }
  }
}

template <int SI, int PM, int NCB>
static void launch_thin_tr_t(const W2Args& a, dim3 grid, hipStream_t s) {
  constexpr int TY = SI == 1 ? 8 : 4;
  constexpr int BZ = 3 * SI + 3, BY = (TY - 1) * SI + 3, BX = 7 * SI + 3;
  constexpr int QBYTES = (BZ * BY * BX * 8 + 63) / 64 * 64;
  size_t lds = QBYTES + (size_t)NCB * (4 * TY * 8) * 64;
  if (lds < 256 * 8 * sizeof(float)) lds = 256 * 8 * sizeof(float);      // the bias-gradient reduction reuses the images
  if (a.q_bf) {
    if constexpr (PM < 2) hipLaunchKernelGGL((wgrad_thin_tr_kernel<SI, PM, NCB, true>), grid, dim3(256), lds, s, a);
  } else {
    hipLaunchKernelGGL((wgrad_thin_tr_kernel<SI, PM, NCB>), grid, dim3(256), lds, s, a);
  }
}

static void launch_thin_tr(const W2Args& a, int si, int ncb, dim3 grid, hipStream_t s) {
  if (a.p_thin) {                                   // both sides thin: stride 1 only (wgeometry)
    if (a.p_bf) launch_thin_tr_t<1, 3, 1>(a, grid, s); else launch_thin_tr_t<1, 2, 1>(a, grid, s);
  } else if (si == 1) {
    if (a.p_bf) { if (ncb == 2) launch_thin_tr_t<1, 1, 2>(a, grid, s); else launch_thin_tr_t<1, 1, 1>(a, grid, s); }
    else { if (ncb == 2) launch_thin_tr_t<1, 0, 2>(a, grid, s); else launch_thin_tr_t<1, 0, 1>(a, grid, s); }
  } else {
    if (a.p_bf) { if (ncb == 2) launch_thin_tr_t<2, 1, 2>(a, grid, s); else launch_thin_tr_t<2, 1, 1>(a, grid, s); }
    else { if (ncb == 2) launch_thin_tr_t<2, 0, 2>(a, grid, s); else launch_thin_tr_t<2, 0, 1>(a, grid, s); }
  }
}

// small_is_cd == 0: dw[(cb*Cs + cs)*ntaps + tap]   (Conv3d with tiny Cin, ConvTranspose3d with tiny Cout)
// small_is_cd == 1: dw[(cs*Cb + cb)*ntaps + tap]   (1x1x1 Conv3d with tiny Cout)
__global__ void wgrad_small_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int nsl, int ntaps, int Cs,
                                          int Cb, int CBp, int small_is_cd, int accumulate, PSets ps) {
  slab += (long long)blockIdx.y * nsl * 128 * CBp;       // parameter set blockIdx.y
  dw += pset_weight_elems(ps, blockIdx.y);
  const int total = ntaps * Cs * Cb;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int cb = i % Cb, row = i / Cb;
    const int tap = row / Cs, cs = row % Cs;
    float s = 0.f;
    for (int sl = 0; sl < nsl; ++sl) s += slab[((long long)sl * 128 + row) * CBp + cb];
    float* o = dw + (small_is_cd ? ((long long)cs * Cb + cb) : ((long long)cb * Cs + cs)) * ntaps + tap;
    *o = accumulate ? (*o + s) : s;
  }
}


// ------------------------------------------------------------------ tiny x tiny weight gradient (VALU)
// Conv3d k3 s1 with <= 4 channels on BOTH sides (the full-resolution R->R unit of the top ResidualUnit): 1 GFLOP
// over 2 M voxels.  On the matrix cores the 3 dense columns would be padded to 32; here a thread owns a voxel,
// keeps the 9 (ky,kx) x CS x CB products of one kz plane in registers and walks rows (a wave = 64 consecutive x
// of one row; blockIdx.y = kz).  x is read as one 16-byte voxel per neighbour (clamped address, the invalid x
// neighbours are dropped by zeroing dy for that kx), norm+ReLU applied on load.  Lanes are combined once at the
// end; workgroup partials land in dw's own layout so the generic row-sum reduce finishes the job.
struct WTArgs {
  TV x; NL tx;
  TV dy;
  float* part;      // [blocks][ld]: dw layout [cb][cs][27]
  float* dbpart;    // [blocks][4] or null
  int ld;
  int B, ips;       // workgroups and batch items per parameter set
  int BT;           // workgroups of the launch per kz plane (B x sets)
};

template <int CS, int CB, bool HAS_T>
__global__ __launch_bounds__(256) void wgrad_tiny_kernel(WTArgs a) {
  __shared__ float red[4][9 * CS * CB + CB];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // The three kz planes of a unit range read the same dy rows and x rows one plane apart.  As grid rows (blockIdx.y = kz)
  // they ran as three sweeps far apart in time: 3x the HBM traffic (PMC r03a: 206 MB per volume against 67 MB of tensors).
  // Now the three workgroups of a range are neighbours in dispatch order ON ONE XCD (ids 8 apart: workgroups are dealt
  // round-robin over the 8 XCDs), so two of them hit that XCD's L2.
  const unsigned xcd = blockIdx.x & 7u, q3 = blockIdx.x >> 3;
  const int kz = (int)(q3 % 3u);
  const unsigned bv = (q3 / 3u) * 8u + xcd;              // this workgroup's id in a launch of BT workgroups
  if (bv >= (unsigned)a.BT) return;
  const int chunks = (a.dy.w + 63) / 64;
  const int hp = (a.dy.h + 1) / 2;                     // a wave takes two adjacent rows per step
  const long long units = (long long)a.ips * a.dy.d * hp * chunks;       // of ONE parameter set
  const int xsw4 = (int)a.x.sw * 4, dsw4 = (int)a.dy.sw * 4;
  float acc[9][CS][CB];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < CS; ++i)
#pragma unroll
      for (int j = 0; j < CB; ++j) acc[t][i][j] = 0.f;
  float dbs[CB];
#pragma unroll
  for (int j = 0; j < CB; ++j) dbs[j] = 0.f;
  int n_cached = -1;
  float sc[CS], sh[CS];
#pragma unroll
  for (int i = 0; i < CS; ++i) { sc[i] = 1.f; sh[i] = 0.f; }
  long long ufirst, ulast;
  const int sx = (int)xcd_contiguous_id(bv, (unsigned)a.BT);
  const int qset = sx / a.B;                            // workgroup sx = workgroup (sx % B) of parameter set (sx / B)
  unit_range(units, sx - qset * a.B, a.B, ufirst, ulast);
  ufirst += (long long)qset * units; ulast += (long long)qset * units;
  // the unit index is decoded ONCE (three 64-bit divisions) and then advanced by carries: all of it wave-uniform
  int chunk, oyp, oz, n;
  {
    long long u = ufirst + wave;
    chunk = (int)(u % chunks); u /= chunks;
    oyp = (int)(u % hp); u /= hp;
    oz = (int)(u % a.dy.d);
    n = (int)(u / a.dy.d);
  }
  auto advance = [&]() {
    chunk += 4;
    while (chunk >= chunks) {
      chunk -= chunks;
      if (++oyp == hp) { oyp = 0; if (++oz == a.dy.d) { oz = 0; ++n; } }
    }
  };
  for (long long u0 = ufirst + wave; u0 < ulast; u0 += 4, advance()) {
    const int oy0 = 2 * oyp;
    const int iz = oz + kz - 1;
    if ((unsigned)iz >= (unsigned)a.x.d) continue;
    if (HAS_T && n != n_cached) {
      nl_coeff_vec<CS>(a.tx, n, CS, 0, sc, sh);
      n_cached = n;
    }
    const int ox = chunk * 64 + lane;
    const bool on = ox < a.dy.w;
    // all 14 loads of the step (2 dy voxels, 4 x rows x 3 neighbours) go out before the first use, from clamped
    // addresses; what does not exist is dropped by zeroing the dy factor
    const char* dyb = reinterpret_cast<const char*>(a.dy.p + (long long)n * a.dy.sn + (long long)oz * a.dy.sd);
    float4 g4[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
      g4[j] = *reinterpret_cast<const float4*>(dyb + (long long)min(oy0 + j, a.dy.h - 1) * a.dy.sh * 4 +
                                               (unsigned)(min(ox, a.dy.w - 1) * dsw4));
    unsigned boff[3];
    bool okx[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int ix = ox + kx - 1;
      okx[kx] = on && (unsigned)ix < (unsigned)a.x.w;
      boff[kx] = (unsigned)(min(max(ix, 0), a.x.w - 1) * xsw4);
    }
    const char* xb = reinterpret_cast<const char*>(a.x.p + (long long)n * a.x.sn + (long long)iz * a.x.sd);
    float4 xr[4][3];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const char* xrow = xb + (long long)min(max(oy0 - 1 + r, 0), a.x.h - 1) * a.x.sh * 4;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) xr[r][kx] = *reinterpret_cast<const float4*>(xrow + boff[kx]);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const bool rowon = oy0 + j < a.dy.h;
      const float gs[4] = {g4[j].x, g4[j].y, g4[j].z, g4[j].w};
      if (kz == 1) {
#pragma unroll
        for (int q = 0; q < CB; ++q) dbs[q] += (on && rowon) ? gs[q] : 0.f;
      }
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int r = j + ky;                       // x row oy0 + j + ky - 1
        const bool rok = rowon && (unsigned)(oy0 - 1 + r) < (unsigned)a.x.h;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const bool ok = rok && okx[kx];
          const float raw[4] = {xr[r][kx].x, xr[r][kx].y, xr[r][kx].z, xr[r][kx].w};
          float g[CB];
#pragma unroll
          for (int q = 0; q < CB; ++q) g[q] = ok ? gs[q] : 0.f;
#pragma unroll
          for (int i = 0; i < CS; ++i) {
            const float xv = HAS_T ? nl_apply(raw[i], sc[i], sh[i], a.tx.relu) : raw[i];
#pragma unroll
            for (int q = 0; q < CB; ++q) acc[ky * 3 + kx][i][q] = fmaf(xv, g[q], acc[ky * 3 + kx][i][q]);
          }
        }
      }
    }
  }
  // lanes -> wave -> workgroup
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < CS; ++i)
#pragma unroll
      for (int j = 0; j < CB; ++j) {
        const float v = wave_sum(acc[t][i][j]);
        if (lane == 0) red[wave][(t * CS + i) * CB + j] = v;
      }
#pragma unroll
  for (int j = 0; j < CB; ++j) {
    const float v = wave_sum(dbs[j]);
    if (lane == 0) red[wave][9 * CS * CB + j] = v;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 9 * CS * CB + CB; e += 256) {
    const float v = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
    if (e < 9 * CS * CB) {
      const int j = e % CB, i = (e / CB) % CS, t = e / (CB * CS);
      a.part[(long long)sx * a.ld + (j * CS + i) * 27 + kz * 9 + t] = v;
    } else if (kz == 1 && a.dbpart != nullptr) {
      a.dbpart[(long long)sx * 4 + (e - 9 * CS * CB)] = v;
    }
  }
}

template <int CS, bool HAS_T>
static void launch_tiny_cb(const WTArgs& a, int cb, int blocks, hipStream_t s) {
  const dim3 grid(3 * ((blocks + 7) / 8) * 8), block(256);      // (plane, workgroup) pairs, see the kernel's id decode
  switch (cb) {
    case 1: hipLaunchKernelGGL((wgrad_tiny_kernel<CS, 1, HAS_T>), grid, block, 0, s, a); break;
    case 2: hipLaunchKernelGGL((wgrad_tiny_kernel<CS, 2, HAS_T>), grid, block, 0, s, a); break;
    case 3: hipLaunchKernelGGL((wgrad_tiny_kernel<CS, 3, HAS_T>), grid, block, 0, s, a); break;
    default: hipLaunchKernelGGL((wgrad_tiny_kernel<CS, 4, HAS_T>), grid, block, 0, s, a); break;
  }
}

template <bool HAS_T>
static void launch_tiny(const WTArgs& a, int cs, int cb, int blocks, hipStream_t s) {
  switch (cs) {
    case 1: launch_tiny_cb<1, HAS_T>(a, cb, blocks, s); break;
    case 2: launch_tiny_cb<2, HAS_T>(a, cb, blocks, s); break;
    case 3: launch_tiny_cb<3, HAS_T>(a, cb, blocks, s); break;
    default: launch_tiny_cb<4, HAS_T>(a, cb, blocks, s); break;
  }
}

// 4-channel vector loads of a tensor are legal: base and every stride keep a quad aligned (16 bytes fp32, 8 bytes bf16)
static bool wvec_ok(const mmtta_tensor* t) {
  return ((uintptr_t)t->ptr) % (is_bf16(t) ? 8 : 16) == 0 && t->sw % 4 == 0 && t->sh % 4 == 0 && t->sd % 4 == 0 && t->sn % 4 == 0;
}

// the transposed-read kernel's loader: 16-byte items of 8 channels (fp32: two quads), 32-bit element offsets built
// from 24-bit multiply-adds
static bool wtr_ok(const mmtta_tensor* t) {
  const int64_t q = is_bf16(t) ? 8 : 4;
  if (((uintptr_t)t->ptr) % 16 != 0 || t->sw % q != 0 || t->sh % q != 0 || t->sd % q != 0 || t->sn % q != 0) return false;
  const int64_t lim24 = (int64_t)1 << 24;
  if (t->sw >= lim24 || t->sh >= lim24 || t->sd >= lim24 || t->d >= lim24 || t->h >= lim24 || t->w >= lim24) return false;
  const int64_t last = (int64_t)(t->d - 1) * t->sd + (int64_t)(t->h - 1) * t->sh + (int64_t)(t->w - 1) * t->sw + roundup(t->c, 8) + 8;
  return last < ((int64_t)1 << 31);
}

struct WGeo {
  bool tiny; int tiny_blocks;
  bool bf16; bool small; int small_is_cd; const mmtta_tensor *q, *pb; bool q_is_x;
  const mmtta_tensor *g, *dn;
  int si, ntaps, TZ, TY, TX;
  int tz, ty, tx, tiles, S, tps, nsl, CGp, CDp;
  int64_t slab_floats, db_floats, colsum_blocks, pre_floats; int pre_chunks;
  bool convt;
  bool tr;        // bf16 27-tap layer on the transposed-read kernel (its own tile shape)
  bool tr1;       // 1x1x1 layer of bf16 precision on the transposed-read streaming kernel
  bool thin_tr;   // thin 27-tap layer of bf16 precision on the transposed-read kernel (wgrad_thin_tr_kernel)
  int ncb;        // its 32-column blocks per workgroup
  bool p_thin;    // ... with <= 4 channels on the dense side as well
  int ips, nsets; // batch items per parameter set, sets per launch: tiles / S / nsl / *_floats / colsum_blocks are PER SET
};

static const int g_wgrad_pair = getenv("MMTTA_WGRAD_PAIR") ? atoi(getenv("MMTTA_WGRAD_PAIR")) : 1;      // (A/B switch: 0 off, 2: every stride-1 layer too)

static int wgeometry(const mmtta_conv_desc* d, const mmtta_tensor* x, const mmtta_tensor* dy, const mmtta_param_sets* sets, WGeo& w) {
  MMTTA_CHECK(d && x && dy && x->ptr && dy->ptr, MMTTA_ERR_INVALID, "wgrad: null argument");
  {
    const int st = psets_validate(sets, x->n);
    if (st) return st;
  }
  // one parameter set = `ips` consecutive batch items; everything below describes ONE set (the launch repeats it nsets times)
  w.ips = sets ? sets->items_per_set : x->n;
  w.nsets = x->n / w.ips;
  MMTTA_CHECK(d->op == MMTTA_CONV_FWD || d->op == MMTTA_CONVT_FWD, MMTTA_ERR_INVALID,
              "wgrad: desc.op must name the module (CONV_FWD or CONVT_FWD)");
  MMTTA_CHECK(d->ksize == 1 || d->ksize == 3, MMTTA_ERR_UNSUPPORTED, "wgrad: ksize %d", d->ksize);
  MMTTA_CHECK(d->stride == 1 || d->stride == 2, MMTTA_ERR_UNSUPPORTED, "wgrad: stride %d", d->stride);
  // desc.dtype selects the MFMA operand type (fp32, or bf16 for the 27-tap layers); accumulation, slabs and the
  // reduced gradient are fp32 either way
  MMTTA_CHECK(d->dtype == MMTTA_F32 || d->dtype == MMTTA_BF16, MMTTA_ERR_UNSUPPORTED, "wgrad: dtype %d", d->dtype);
  MMTTA_CHECK(is_cl(x) && is_cl(dy), MMTTA_ERR_UNSUPPORTED, "wgrad: tensors must be channels-last");
  MMTTA_CHECK(x->c == d->cin && dy->c == d->cout && x->n == dy->n, MMTTA_ERR_INVALID, "wgrad: channel/batch mismatch");
  w.convt = d->op == MMTTA_CONVT_FWD;
  w.tr = false;
  w.tr1 = false;
  w.thin_tr = false;
  w.ncb = 1;
  w.p_thin = false;
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
  // both sides <= 4 channels, k3 s1, 16-byte voxel rows: VALU kernel, partials in dw layout
  auto al16 = [](const mmtta_tensor* t) {
    return ((uintptr_t)t->ptr) % 16 == 0 && t->sc == 1 && t->sw % 4 == 0 && t->sh % 4 == 0 && t->sd % 4 == 0 && t->sn % 4 == 0 &&
           (long long)t->w * t->sw * 4 < (1LL << 31);
  };
  // fp32 voxels of 16 bytes addressed by 32-bit element offsets inside a batch item (wgrad_thin_tr_kernel's thin operands)
  auto q_ok = [](const mmtta_tensor* t) {
    const int64_t last = (int64_t)(t->d - 1) * t->sd + (int64_t)(t->h - 1) * t->sh + (int64_t)(t->w - 1) * t->sw + 8;
    return is_f32(t) && ((uintptr_t)t->ptr) % 16 == 0 && t->sw % 4 == 0 && t->sh % 4 == 0 && t->sd % 4 == 0 && t->sn % 4 == 0 &&
           last < ((int64_t)1 << 31);
  };
  auto q_ok16 = [](const mmtta_tensor* t) {          // ... or bf16 voxels of 8 bytes
    const int64_t last = (int64_t)(t->d - 1) * t->sd + (int64_t)(t->h - 1) * t->sh + (int64_t)(t->w - 1) * t->sw + 8;
    return is_bf16(t) && ((uintptr_t)t->ptr) % 8 == 0 && t->sw % 4 == 0 && t->sh % 4 == 0 && t->sd % 4 == 0 && t->sn % 4 == 0 &&
           last < ((int64_t)1 << 31);
  };
  w.tiny = !w.convt && d->cin <= 4 && d->cout <= 4 && d->ksize == 3 && d->stride == 1 && al16(x) && al16(dy);
  // bf16 precision with the thin layers on the matrix cores (MMTTA_OPT_THIN_MFMA, like their forward / input gradient):
  // the transposed-read kernel below instead of the fp32 vector-ALU kernel
  if (w.tiny && d->dtype == MMTTA_BF16 && g_thin_mfma && g_wgrad_vec && q_ok(x) && (q_ok(dy) || q_ok16(dy))) w.tiny = false;
  if (w.tiny) {
    const long long units = (long long)w.ips * dy->d * ((dy->h + 1) / 2) * ((dy->w + 63) / 64);
    long long blocks = (units + 3) / 4;
    if (blocks > 512) blocks = 512;
    w.tiny_blocks = (int)blocks;
    w.small = false; w.bf16 = false; w.small_is_cd = 0; w.q = w.pb = nullptr; w.q_is_x = false;
    w.TZ = w.TY = w.TX = 0; w.tz = w.ty = w.tx = w.tiles = w.S = w.tps = w.nsl = 0; w.CGp = w.CDp = 0;
    w.slab_floats = (int64_t)blocks * (27 * d->cin * d->cout);
    w.db_floats = (int64_t)blocks * 4;
    w.pre_floats = 0; w.pre_chunks = 0; w.colsum_blocks = 0;
    return MMTTA_OK;
  }
  // small-channel path: which tensor is the small gathered one (Q) and which the dense one (P)
  w.small = false; w.bf16 = false; w.small_is_cd = 0; w.q = w.pb = nullptr; w.q_is_x = false;
  if (!w.convt && d->cin <= 4) { w.small = true; w.q = x; w.pb = dy; w.q_is_x = true; }
  else if (!w.convt && d->cout <= 4 && d->ksize == 1 &&
           !(d->dtype == MMTTA_BF16 && d->cin >= 16 && wtr_ok(x) && wtr_ok(dy) && (!is_bf16(dy) || is_bf16(x)) && g_wgrad_vec)) {
    // (in bf16 precision the 1x1x1 streaming kernel below takes these heads too: N padded to 32 costs nothing there)
    w.small = true; w.q = dy; w.pb = x; w.small_is_cd = 1;
  }
  else if (w.convt && d->cout <= 4) { w.small = true; w.q = dy; w.pb = x; }
  if (w.small) {
    w.CGp = 128;
    w.CDp = roundup(w.pb->c, 32);
    // bf16 precision, 27 taps: the transposed-read kernel when the operands admit its staging (Q: fp32 voxels of 16 bytes;
    // P: 16-byte items of 8 channels; 32-bit element offsets inside a batch item)
    w.p_thin = w.pb->c <= 4 && w.si == 1 && g_thin_mfma;
    // (a bf16-stored thin tensor - the network input of bf16 precision, a thin gradient - has 8-byte voxels: this kernel only)
    w.thin_tr = d->dtype == MMTTA_BF16 && w.ntaps == 27 && g_wgrad_vec &&
                (q_ok(w.q) || (q_ok16(w.q) && !w.p_thin)) &&
                (w.p_thin ? (q_ok(w.pb) || q_ok16(w.pb)) : (wtr_ok(w.pb) && w.pb->c % 8 == 0));
    if (!w.thin_tr) w.p_thin = false;
    w.ncb = (w.thin_tr && (w.CDp / 32) % 2 == 0) ? 2 : 1;
    w.TZ = 4; w.TY = (w.thin_tr && w.si == 1) ? 8 : 4; w.TX = 8;
    w.tz = (w.pb->d + 3) / 4; w.ty = (w.pb->h + w.TY - 1) / w.TY; w.tx = (w.pb->w + 7) / 8;
    w.tiles = w.tz * w.ty * w.tx * w.ips;
    // slabs: one volume in flight 256 / 512 / 768 / 1024 -> 56 / 46 / 57 / 57 us per launch at 128^3; two in flight
    // (method.lanes: 2, the default) 256 edges out 512 for the whole step (41.9 vs 41.5 volumes/s): less slab traffic
    int S = g_tune[3] / (w.CDp / 32 / w.ncb);
    // the transposed-read kernel does little per tile (8 MFMAs a wave behind one round of loads): it hides its memory latency
    // with workgroups per CU (86 - 172 registers: 3 - 5 per SIMD), so it wants 4x the slabs, down to 8 tiles each
    // (24 volumes in flight, per group of 8, 42 -> 168 slabs per volume: 3 -> 3 at 128^3 475 -> 265 us, 4 -> 32 145 -> 123, the
    // up-convolution 262 -> 239; 336: 238 / 128 / 241; 672 slabs: 249 / 142 / 272)
    if (w.thin_tr) S = std::min(4 * S, std::max(1, w.tiles / 8));
    if (S < 1) S = 1;
    if (S > w.tiles) S = w.tiles;
    w.tps = (w.tiles + S - 1) / S;
    w.S = (w.tiles + w.tps - 1) / w.tps;
    w.nsl = w.S;
    w.slab_floats = (int64_t)w.nsl * 128 * w.CDp;
    w.pre_chunks = w.nsl > 32 ? (w.nsl + 31) / 32 : 0;
    w.pre_floats = (int64_t)w.pre_chunks * 128 * w.CDp;
    w.colsum_blocks = 0;
    const bool bias_from_p = (w.pb == dy);           // P carries the output channels
    if (bias_from_p) w.db_floats = (int64_t)w.nsl * w.CDp;
    else { w.colsum_blocks = (int64_t)w.ips * channel_partial_rows(dy); w.db_floats = w.colsum_blocks * 2 * dy->c; }
    return MMTTA_OK;
  }
  // bf16 operands: the transposed-read kernel, for operand pairs that admit its 16-byte items; anything else (ragged
  // channel slices, MMTTA_OPT_WGRAD_VECTOR_STAGING = 0) computes on the fp32-operand kernel
  // (a gradient may be bf16-stored only next to a bf16-stored module input: x is the gathered operand of a convolution, the
  // dense one of a transposed convolution)
  const bool grad_bf_ok = !is_bf16(dy) || is_bf16(x);
  w.tr = d->dtype == MMTTA_BF16 && w.ntaps == 27 && wtr_ok(w.g) && wtr_ok(w.dn) && grad_bf_ok && g_wgrad_vec;
  w.bf16 = w.tr;
  w.tr1 = !w.convt && d->dtype == MMTTA_BF16 && w.ntaps == 1 && w.si == 1 && wtr_ok(w.g) && wtr_ok(w.dn) && grad_bf_ok &&
          g_wgrad_vec;
  if ((w.tr && w.si == 1) || w.tr1) { w.TZ = 4; w.TY = 8; w.TX = 8; }
  else if (w.si == 1) { w.TZ = 4; w.TY = 4; w.TX = 8; }
  else if (w.tr) { w.TZ = 2; w.TY = 4; w.TX = 8; }
  else { w.TZ = 2; w.TY = 2; w.TX = 8; }
  w.tz = (w.dn->d + w.TZ - 1) / w.TZ;
  w.ty = (w.dn->h + w.TY - 1) / w.TY;
  w.tx = (w.dn->w + w.TX - 1) / w.TX;
  w.tiles = w.tz * w.ty * w.tx * w.ips;
  w.CGp = roundup(w.g->c, 32);
  w.CDp = roundup(w.dn->c, 32);
  // both operands bf16-stored: a workgroup takes two dense column blocks per staged box (wgrad_tr_kernel NB) - every
  // stride-2 layer, and the stride-1 layers of <= 128 dense channels (per group of 8 volumes: conv 32->64 s2 162 -> 112 us,
  // convT 128->32 307 -> 210, convT 256->64 200 -> 144, 64->128 s2 102 -> 64; 64->64 s1 120 -> 98, 128->128 83 -> 73; the
  // 256- and 512-channel stride-1 layers have hundreds of workgroups and want two of them per CU: 512->512 218 -> 224)
  if (w.tr && g_wgrad_pair && (w.si == 2 || w.CDp <= 128 || g_wgrad_pair >= 2) && is_bf16(w.g) && is_bf16(w.dn) && (w.CDp / 32) % 2 == 0)
    w.ncb = 2;
  // (a pair counts as one block: the launch keeps its workgroup count and takes twice the slabs - measured against the same
  // slabs with half the workgroups, three runs each on one box: 96.5 / 96.1 volumes/s, 94.7 without pairs)
  const int blocks_cc = (w.CGp / 32) * (w.CDp / 32 / w.ncb);
  // workgroups (= slabs x channel blocks) per launch.  One volume in flight: 512 beats 256 by 3 % of the weight-gradient
  // time; two in flight (method.lanes: 2, the default) the other lane fills the CUs and halving the slab traffic wins:
  // 512 / 256 / 128 -> 40.1 / 41.5 / 40.6 volumes/s
  int S = g_tune[2] / blocks_cc;
  if (w.tr1) S = 1024 / blocks_cc;       // a streaming kernel: enough workgroups to keep HBM busy (slabs are 4 KB each)
  if (S < 1) S = 1;
  if (S > w.tiles) S = w.tiles;
  w.tps = (w.tiles + S - 1) / S;
  w.S = (w.tiles + w.tps - 1) / w.tps;
  w.nsl = w.S * (w.ntaps == 1 ? 4 : 1);
  w.slab_floats = (int64_t)w.nsl * w.ntaps * w.CGp * w.CDp;
  w.pre_chunks = w.nsl > 32 ? (w.nsl + 31) / 32 : 0;
  w.pre_floats = (int64_t)w.pre_chunks * w.ntaps * w.CGp * w.CDp;
  w.colsum_blocks = 0;
  if (w.convt) {
    w.colsum_blocks = (int64_t)w.ips * channel_partial_rows(dy);
    w.db_floats = w.colsum_blocks * 2 * dy->c;
  } else {
    w.db_floats = (int64_t)w.nsl * w.CDp;
  }
  return MMTTA_OK;
}

template <int TZ, int TY, int TX, int NTW, bool GBF, bool DBF>
static int launch_wgrad_t(const WArgs& a, int S, hipStream_t s) {
  const int ext = a.ntaps == 1 ? 0 : 2;
  const int BZ = (TZ - 1) * a.si + ext + 1, BY = (TY - 1) * a.si + ext + 1, BX = (TX - 1) * a.si + ext + 1;
  const size_t lds = ((size_t)BZ * BY * BX + TZ * TY * TX) * 32 * sizeof(float);
  auto kern = wgrad_f32_kernel<TZ, TY, TX, NTW, GBF, DBF>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  dim3 grid(S, a.CGp / 32, a.CDp / 32);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
  return launch_status("conv wgrad");
}

template <int TZ, int TY, int TX, int NTW>
static int launch_wgrad(const WArgs& a, int S, hipStream_t s) {
  if (a.g_bf && a.d_bf) return launch_wgrad_t<TZ, TY, TX, NTW, true, true>(a, S, s);
  if (a.g_bf) return launch_wgrad_t<TZ, TY, TX, NTW, true, false>(a, S, s);
  if (a.d_bf) return launch_wgrad_t<TZ, TY, TX, NTW, false, true>(a, S, s);
  return launch_wgrad_t<TZ, TY, TX, NTW, false, false>(a, S, s);
}

}  // namespace mmtta

using namespace mmtta;

extern "C" int64_t mmtta_conv_wgrad_workspace_bytes_sets(const mmtta_conv_desc* d, const mmtta_tensor* x, const mmtta_tensor* dy,
                                                         const mmtta_param_sets* sets) {
  WGeo w;
  if (wgeometry(d, x, dy, sets, w)) return -1;
  return (w.slab_floats + w.db_floats + w.pre_floats) * w.nsets * (int64_t)sizeof(float);
}

extern "C" int64_t mmtta_conv_wgrad_workspace_bytes(const mmtta_conv_desc* d, const mmtta_tensor* x,
                                                    const mmtta_tensor* dy) {
  return mmtta_conv_wgrad_workspace_bytes_sets(d, x, dy, nullptr);
}

extern "C" int mmtta_conv_wgrad_kernel(const mmtta_conv_desc* d, const mmtta_tensor* x, const mmtta_tensor* dy) {
  WGeo w;
  const int st = wgeometry(d, x, dy, nullptr, w);
  if (st) return st < 0 ? st : -st;
  if (w.tiny) return 6;
  if (w.small) return w.thin_tr ? 10 : 3;
  if (w.tr1) return 9;
  if (w.tr) return w.si == 1 ? 7 : 8;
  if (w.ntaps == 1) return 2;
  return w.si == 1 ? 0 : 1;
}

static int conv_wgrad_impl(const mmtta_conv_desc* d, const mmtta_tensor* x, const mmtta_norm_on_load* x_norm,
                           const mmtta_tensor* dy, float* dw, float* db, int accumulate, void* workspace,
                           int64_t workspace_bytes, const mmtta_param_sets* sets, void* stream) {
  WGeo w;
  int st = wgeometry(d, x, dy, sets, w);
  if (st) return st;
  const PSets ps = psets(sets);
  const int Q = w.nsets;                 // the workspace regions below hold Q sets back to back (set-major)
  MMTTA_CHECK(dw != nullptr, MMTTA_ERR_INVALID, "wgrad: null dw");
  const int64_t need = (w.slab_floats + w.db_floats + w.pre_floats) * Q * 4;
  MMTTA_CHECK(workspace != nullptr && workspace_bytes >= need, MMTTA_ERR_WORKSPACE, "wgrad: workspace %lld bytes, need %lld",
              (long long)workspace_bytes, (long long)need);
  hipStream_t s = (hipStream_t)stream;
  if (w.tiny) {
    MMTTA_CHECK(is_f32(x) && is_f32(dy), MMTTA_ERR_UNSUPPORTED, "wgrad (tiny layer): fp32-stored tensors only");
    WTArgs t;
    t.x = tv(x); t.tx = nl(x_norm); t.dy = tv(dy);
    t.part = (float*)workspace; t.ld = 27 * d->cin * d->cout;
    t.dbpart = db != nullptr ? (float*)workspace + w.slab_floats * Q : nullptr;
    t.B = w.tiny_blocks; t.ips = w.ips; t.BT = w.tiny_blocks * Q;
    if (t.tx.mean != nullptr || t.tx.scale != nullptr) launch_tiny<true>(t, d->cin, d->cout, w.tiny_blocks * Q, s);
    else launch_tiny<false>(t, d->cin, d->cout, w.tiny_blocks * Q, s);
    st = launch_status("wgrad tiny");
    if (st || g_profile_main_only) return st;
    hipLaunchKernelGGL(db_reduce_kernel, dim3(t.ld, Q), dim3(64), 0, s, t.part, dw, w.tiny_blocks, t.ld, t.ld, accumulate, ps, 1);
    if (db != nullptr)
      hipLaunchKernelGGL(db_reduce_kernel, dim3(d->cout, Q), dim3(64), 0, s, t.dbpart, db, w.tiny_blocks, d->cout, 4, accumulate, ps, 0);
    return launch_status("wgrad tiny reduce");
  }
  if (w.small) {
    W2Args b;
    b.q = (const float*)w.q->ptr; b.qsn = w.q->sn; b.qsd = w.q->sd; b.qsh = w.q->sh; b.qsw = w.q->sw;
    b.Cs = w.q->c; b.Dq = w.q->d; b.Hq = w.q->h; b.Wq = w.q->w;
    b.p = (const float*)w.pb->ptr; b.psn = w.pb->sn; b.psd = w.pb->sd; b.psh = w.pb->sh; b.psw = w.pb->sw;
    b.Cb = w.pb->c; b.Dp = w.pb->d; b.Hp = w.pb->h; b.Wp = w.pb->w;
    b.tq = w.q_is_x ? nl(x_norm) : nl(nullptr);
    b.tp = w.q_is_x ? nl(nullptr) : nl(x_norm);
    b.si = w.si; b.ntaps = w.ntaps;
    b.slab = (float*)workspace;
    float* dbws2 = (float*)workspace + w.slab_floats * Q;
    const bool bias_from_p = (w.pb == dy);
    b.dbpart = (db != nullptr && bias_from_p) ? dbws2 : nullptr;
    b.tz = w.tz; b.ty = w.ty; b.tx = w.tx; b.tiles = w.tiles * Q; b.tiles_per_split = w.tps; b.CBp = w.CDp;
    b.S = w.S; b.tiles_set = w.tiles;
    auto al4 = [](const mmtta_tensor* t) {
      return ((((uintptr_t)t->ptr) % 16 == 0) && t->sw % 4 == 0 && t->sh % 4 == 0 && t->sd % 4 == 0 && t->sn % 4 == 0) ? 1 : 0;
    };
    b.qvec4 = al4(w.q); b.pvec4 = al4(w.pb);
    MMTTA_CHECK(is_f32(w.q) || w.thin_tr, MMTTA_ERR_UNSUPPORTED, "wgrad (thin layer): the <= 4-channel tensor must be fp32-stored");
    b.p_bf = is_bf16(w.pb) ? 1 : 0;
    if (b.p_bf) b.pvec4 = (((uintptr_t)w.pb->ptr) % 8 == 0 && w.pb->sw % 4 == 0 && w.pb->sh % 4 == 0 && w.pb->sd % 4 == 0 && w.pb->sn % 4 == 0) ? 1 : 0;
    b.bf = (d->dtype == MMTTA_BF16 && w.ntaps == 27) ? 1 : 0;
    b.p_thin = w.p_thin ? 1 : 0;
    b.q_bf = is_bf16(w.q) ? 1 : 0;
    MMTTA_CHECK(!b.q_bf || w.thin_tr, MMTTA_ERR_UNSUPPORTED, "wgrad (thin layer): a bf16-stored <= 4-channel tensor needs the transposed-read kernel");
    const int ext = w.ntaps == 1 ? 0 : 2;
    const int BZ = 3 * w.si + ext + 1, BY = 3 * w.si + ext + 1, BX = 7 * w.si + ext + 1;
    const size_t lds = ((size_t)BZ * BY * BX * 4 + 128 * 32) * sizeof(float);
    MMTTA_CHECK((w.ntaps == 27 && (w.si == 1 || w.si == 2)) || (w.ntaps == 1 && w.si == 1), MMTTA_ERR_UNSUPPORTED,
                "wgrad (thin layer): %d taps with stride %d", w.ntaps, w.si);
    const dim3 sg(w.S * Q, w.CDp / 32);
    if (w.thin_tr) {
      launch_thin_tr(b, w.si, w.ncb, dim3(w.S * Q, w.CDp / 32 / w.ncb), s);
    } else if (w.ntaps == 1) {
      if (b.p_bf) hipLaunchKernelGGL((wgrad_small_kernel<true, 1, false>), sg, dim3(256), lds, s, b);
      else hipLaunchKernelGGL((wgrad_small_kernel<false, 1, false>), sg, dim3(256), lds, s, b);
    } else if (w.si == 1) {
      if (b.p_bf) hipLaunchKernelGGL((wgrad_small_kernel<true, 1, true>), sg, dim3(256), lds, s, b);
      else hipLaunchKernelGGL((wgrad_small_kernel<false, 1, true>), sg, dim3(256), lds, s, b);
    } else {
      if (b.p_bf) hipLaunchKernelGGL((wgrad_small_kernel<true, 2, true>), sg, dim3(256), lds, s, b);
      else hipLaunchKernelGGL((wgrad_small_kernel<false, 2, true>), sg, dim3(256), lds, s, b);
    }
    st = launch_status("wgrad small");
    if (st || g_profile_main_only) return st;
    const int total = w.ntaps * b.Cs * b.Cb;
    const float* rsrc = b.slab;
    int rn = w.nsl;
    if (w.pre_chunks > 0) {
      float* pre = (float*)workspace + (w.slab_floats + w.db_floats) * Q;
      const long long elems = (long long)128 * w.CDp;
      hipLaunchKernelGGL(slab_prereduce_kernel, dim3((unsigned)((elems + 255) / 256), w.pre_chunks, Q), dim3(256), 0, s, b.slab,
                         pre, w.nsl, elems);
      st = launch_status("wgrad small prereduce");
      if (st) return st;
      rsrc = pre; rn = w.pre_chunks;
    }
    hipLaunchKernelGGL(wgrad_small_reduce_kernel, dim3((total + 255) / 256, Q), dim3(256), 0, s, rsrc, dw, rn, w.ntaps, b.Cs,
                       b.Cb, w.CDp, w.small_is_cd, accumulate, ps);
    st = launch_status("wgrad small reduce");
    if (st) return st;
    if (db != nullptr) {
      if (bias_from_p) {
        hipLaunchKernelGGL(db_reduce_kernel, dim3(b.Cb, Q), dim3(64), 0, s, dbws2, db, w.nsl, b.Cb, w.CDp, accumulate, ps, 0);
      } else {
        st = launch_channel_sums(dy, dbws2, s);          // rows are n-major: a set's rows are contiguous
        if (st) return st;
        hipLaunchKernelGGL(db_reduce_kernel, dim3(dy->c, Q), dim3(64), 0, s, dbws2, db, (int)w.colsum_blocks, dy->c, 2 * dy->c,
                           accumulate, ps, 0);
      }
      st = launch_status("bias reduce");
    }
    return st;
  }
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
  float* dbws = (float*)workspace + w.slab_floats * Q;
  a.dbpart = (db != nullptr && !w.convt) ? dbws : nullptr;
  a.tz = w.tz; a.ty = w.ty; a.tx = w.tx; a.tiles = w.tiles * Q; a.tiles_per_split = w.tps;
  a.S = w.S; a.tiles_set = w.tiles;
  a.CGp = w.CGp; a.CDp = w.CDp;
  a.g_bf = is_bf16(w.g) ? 1 : 0;
  a.d_bf = is_bf16(w.dn) ? 1 : 0;
  a.convt = w.convt ? 1 : 0;
  a.gvec4 = wvec_ok(w.g) ? 1 : 0;
  a.dvec4 = wvec_ok(w.dn) ? 1 : 0;
  const int SQ = w.S * Q;                // slabs of the launch
  if (w.tr1) st = launch_wgrad_tr1(a, SQ, s);
  else if (w.tr) st = (w.si == 1) ? launch_wgrad_tr<4, 8, 1>(a, SQ, s, w.ncb) : launch_wgrad_tr<2, 4, 2>(a, SQ, s, w.ncb);
  else if (w.ntaps == 1) st = launch_wgrad<4, 4, 8, 1>(a, SQ, s);
  else st = (w.si == 1) ? launch_wgrad<4, 4, 8, 7>(a, SQ, s) : launch_wgrad<2, 2, 8, 7>(a, SQ, s);
  if (st || g_profile_main_only) return st;
  const float* rsrc = a.slab;
  int rn = w.nsl;
  if (w.pre_chunks > 0) {
    float* pre = (float*)workspace + (w.slab_floats + w.db_floats) * Q;
    const long long elems = (long long)w.ntaps * w.CGp * w.CDp;
    hipLaunchKernelGGL(slab_prereduce_kernel, dim3((unsigned)((elems + 255) / 256), w.pre_chunks, Q), dim3(256), 0, s, a.slab, pre,
                       w.nsl, elems);
    st = launch_status("wgrad prereduce");
    if (st) return st;
    rsrc = pre; rn = w.pre_chunks;
  }
  const bool db_here = db != nullptr && !w.convt;      // bias partials written by the main kernel: [nsl][CDp]
  if (w.ntaps == 27)
    hipLaunchKernelGGL(wgrad_reduce27_kernel, dim3(a.Cg + (db_here ? 1 : 0), (a.Cd + 31) / 32, Q), dim3(256), 0, s, rsrc, dw, rn,
                       a.Cg, a.Cd, w.CGp, w.CDp, accumulate, dbws, db_here ? db : nullptr, w.nsl, ps);
  else
    hipLaunchKernelGGL(wgrad_reduce1_kernel, dim3((a.Cg + 31) / 32, (a.Cd + 31) / 32, Q), dim3(256), 0, s, rsrc, dw, rn,
                       a.Cg, a.Cd, w.CGp, w.CDp, accumulate, ps);
  st = launch_status("wgrad reduce");
  if (st) return st;
  if (db != nullptr) {
    if (!w.convt) {
      if (w.ntaps != 27)
        hipLaunchKernelGGL(db_reduce_kernel, dim3(a.Cd, Q), dim3(64), 0, s, dbws, db, w.nsl, a.Cd, w.CDp, accumulate, ps, 0);
    } else {
      // ConvTranspose3d bias gradient = per-channel sum of dy over the fine grid
      st = launch_channel_sums(dy, dbws, s);
      if (st) return st;
      hipLaunchKernelGGL(db_reduce_kernel, dim3(dy->c, Q), dim3(64), 0, s, dbws, db, (int)w.colsum_blocks, dy->c,
                         2 * dy->c, accumulate, ps, 0);
    }
    st = launch_status("bias reduce");
  }
  return st;
}

extern "C" int mmtta_conv_wgrad(const mmtta_conv_desc* d, const mmtta_tensor* x, const mmtta_norm_on_load* x_norm,
                                const mmtta_tensor* dy, float* dw, float* db, int accumulate, void* workspace,
                                int64_t workspace_bytes, void* stream) {
  return conv_wgrad_impl(d, x, x_norm, dy, dw, db, accumulate, workspace, workspace_bytes, nullptr, stream);
}

extern "C" int mmtta_conv_wgrad_sets(const mmtta_conv_desc* d, const mmtta_tensor* x, const mmtta_norm_on_load* x_norm,
                                     const mmtta_tensor* dy, float* dw, float* db, int accumulate, void* workspace,
                                     int64_t workspace_bytes, const mmtta_param_sets* sets, void* stream) {
  return conv_wgrad_impl(d, x, x_norm, dy, dw, db, accumulate, workspace, workspace_bytes, sets, stream);
}
