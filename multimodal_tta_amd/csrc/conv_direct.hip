// Direct convolution for layers that PRODUCE at most 4 channels (gfx950).
//
// The U-Net's last layers run at full resolution with 3 (BraTS) or 1 (HECKTOR) output channels
// (reference src/models/unet.py:56-66 -> monai UNet top level: ConvTranspose3d 64->R and the
// conv-only ResidualUnit R->R; src/models/unet_multimodal_midfusion.py:196 final_conv 32->R).
// On the matrix cores their N dimension would be padded 3 -> 32 (10x wasted MFMA work on the largest
// grids), while their real arithmetic is tiny: they are HBM / L1 bound.  So: one thread per output voxel
// (lanes along W, coalesced), weights [tap][K][4] broadcast from LDS, the norm-on-load coefficients of the
// input in LDS, fp32 FMA accumulation in registers, the same epilogue as the implicit-GEMM kernel
// (bias, fused residual add, per-block statistics for a following norm).
//
// Two gather modes cover all four ops:
//   gather      in = out*stride + k - pad                    (Conv3d forward, ConvTranspose3d input gradient)
//   transposed  t = out + pad - k, in = t/stride if t % stride == 0   (ConvTranspose3d forward, Conv3d input gradient)
#include "common.h"

namespace mmtta {

typedef float float2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 ubf16x8 __attribute__((ext_vector_type(8)));
typedef float ufloat16 __attribute__((ext_vector_type(16)));

struct DArgs {
  TV in; NL tin;
  TV out;
  const float* w;      // [T][K][4]
  const float* bias;   // [N] or null
  const float* add; long long asn, asd, ash, asw; NL tadd;
  int K, N, ksize, stride, transposed, accumulate;
  float* stats; int blocks_per_n;
  PSets ps;            // per-item parameter sets: w / bias are set 0's (common.h)
  int out_vec4;        // N < 4, 16-byte voxel rows whose pad lanes this view owns: one full store per voxel
};

// valid taps of one axis for output coordinate o: (k, input coordinate) pairs
__device__ __forceinline__ int axis_taps(int o, int ksize, int stride, int transposed, int in_extent, int* ks, int* is) {
  const int pad = (ksize - 1) / 2;
  int cnt = 0;
  for (int k = 0; k < ksize; ++k) {
    int i;
    if (!transposed) i = o * stride + k - pad;
    else {
      const int t = o + pad - k;
      if (t < 0 || (stride == 2 && (t & 1))) continue;
      i = stride == 2 ? (t >> 1) : t;
    }
    if ((unsigned)i < (unsigned)in_extent) { ks[cnt] = k; is[cnt] = i; ++cnt; }
  }
  return cnt;
}

template <bool HAS_T>
__global__ __launch_bounds__(256) void direct_conv_kernel(DArgs a) {
  extern __shared__ float lds[];
  const int T = a.ksize * a.ksize * a.ksize;
  float* wl = lds;                       // [T*K][4]
  float* coef = lds + T * a.K * 4;       // [K][2] scale, shift
  float* red = coef + 2 * a.K;           // [sum|sq][wave][channel] = 32 floats
  const int n = blockIdx.y;
  a.w = pset_packed(a.ps, a.w, n); a.bias = pset_bias(a.ps, a.bias, n);      // this batch item's parameter set
  for (int i = threadIdx.x; i < T * a.K * 4; i += 256) wl[i] = a.w[i];
  if (HAS_T)
    for (int k = threadIdx.x; k < a.K; k += 256) {
      float sc, sh;
      nl_coeff(a.tin, n, a.K, k, sc, sh);
      coef[2 * k] = sc; coef[2 * k + 1] = sh;
    }
  __syncthreads();
  const long long dhw = (long long)a.out.d * a.out.h * a.out.w;
  const long long v = (long long)blockIdx.x * 256 + threadIdx.x;
  const bool active = v < dhw;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  int ox = 0, oy = 0, oz = 0;
  if (active) {
    long long t = v;
    ox = (int)(t % a.out.w); t /= a.out.w;
    oy = (int)(t % a.out.h);
    oz = (int)(t / a.out.h);
    const float* inb = a.in.p + (long long)n * a.in.sn;
    const bool vec = (a.K % 4 == 0) && (a.in.sw % 4 == 0) && (a.in.sh % 4 == 0) && (a.in.sd % 4 == 0) &&
                     (a.in.sn % 4 == 0) && (((uintptr_t)a.in.p) % 16 == 0);
    int kzs[3], izs[3], kys[3], iys[3], kxs[3], ixs[3];
    const int nz = axis_taps(oz, a.ksize, a.stride, a.transposed, a.in.d, kzs, izs);
    const int ny = axis_taps(oy, a.ksize, a.stride, a.transposed, a.in.h, kys, iys);
    const int nx = axis_taps(ox, a.ksize, a.stride, a.transposed, a.in.w, kxs, ixs);
    for (int az = 0; az < nz; ++az)
      for (int ay = 0; ay < ny; ++ay)
        for (int ax = 0; ax < nx; ++ax) {
          const float* ip = inb + (long long)izs[az] * a.in.sd + (long long)iys[ay] * a.in.sh + (long long)ixs[ax] * a.in.sw;
          const float* wt = wl + ((kzs[az] * a.ksize + kys[ay]) * a.ksize + kxs[ax]) * a.K * 4;
          if (vec) {
            for (int k = 0; k < a.K; k += 4) {
              const float4 x4 = *reinterpret_cast<const float4*>(ip + k);
              const float xs[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const float xv = HAS_T ? nl_apply(xs[j], coef[2 * (k + j)], coef[2 * (k + j) + 1], a.tin.relu) : xs[j];
                const float4 w4 = *reinterpret_cast<const float4*>(wt + (k + j) * 4);
                acc[0] = fmaf(xv, w4.x, acc[0]); acc[1] = fmaf(xv, w4.y, acc[1]);
                acc[2] = fmaf(xv, w4.z, acc[2]); acc[3] = fmaf(xv, w4.w, acc[3]);
              }
            }
          } else {
            for (int k = 0; k < a.K; ++k) {
              const float xv = HAS_T ? nl_apply(ip[k], coef[2 * k], coef[2 * k + 1], a.tin.relu) : ip[k];
              const float4 w4 = *reinterpret_cast<const float4*>(wt + k * 4);
              acc[0] = fmaf(xv, w4.x, acc[0]); acc[1] = fmaf(xv, w4.y, acc[1]);
              acc[2] = fmaf(xv, w4.z, acc[2]); acc[3] = fmaf(xv, w4.w, acc[3]);
            }
          }
        }
  }
  // ---- epilogue
  float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
  if (active) {
    float* op = a.out.p + (long long)n * a.out.sn + (long long)oz * a.out.sd + (long long)oy * a.out.sh +
                (long long)ox * a.out.sw;
    const float* ap = a.add ? a.add + (long long)n * a.asn + (long long)oz * a.asd + (long long)oy * a.ash +
                                  (long long)ox * a.asw : nullptr;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (c < a.N) {
        float val = acc[c] + (a.bias ? a.bias[c] : 0.f);
        if (ap) {
          float sc, sh;
          nl_coeff(a.tadd, n, a.N, c, sc, sh);
          val += nl_apply(ap[c], sc, sh, a.tadd.relu);
        }
        if (a.accumulate) val += op[c];
        op[c] = val;
        ssum[c] = val; ssq[c] = val * val;
      }
    }
  }
  if (a.stats != nullptr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float s = wave_sum(ssum[c]), q = wave_sum(ssq[c]);
      if (lane == 0) { red[(0 * 4 + wave) * 4 + c] = s; red[(1 * 4 + wave) * 4 + c] = q; }
    }
    __syncthreads();
    if (threadIdx.x < a.N) {
      const int c = threadIdx.x;
      const float s = red[0 * 4 + c] + red[1 * 4 + c] + red[2 * 4 + c] + red[3 * 4 + c];
      const float q = red[16 + 0 * 4 + c] + red[16 + 1 * 4 + c] + red[16 + 2 * 4 + c] + red[16 + 3 * 4 + c];
      const long long row = (long long)n * a.blocks_per_n + blockIdx.x;
      a.stats[(row * 2 + 0) * a.N + c] = s;
      a.stats[(row * 2 + 1) * a.N + c] = q;
    }
  }
}

// ------------------------------------------------------------------ shared epilogue
// One lane owns one output voxel with NO channel values: bias, fused residual add, accumulate, store; the
// values are also added into the lane's running statistics.
template <int NO>
__device__ __forceinline__ void direct_epilogue(const DArgs& a, int n, int oz, int oy, int ox, const float* val_in,
                                                float* ssum, float* ssq) {
  float* op = a.out.p + (long long)n * a.out.sn + (long long)oz * a.out.sd + (long long)oy * a.out.sh +
              (long long)ox * a.out.sw;
  const float* ap = a.add ? a.add + (long long)n * a.asn + (long long)oz * a.asd + (long long)oy * a.ash +
                                (long long)ox * a.asw : nullptr;
  float vals[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < NO; ++c) {
    float val = val_in[c] + (a.bias ? a.bias[c] : 0.f);
    if (ap) {
      float s1, h1;
      nl_coeff(a.tadd, n, a.N, c, s1, h1);
      val += nl_apply(ap[c], s1, h1, a.tadd.relu);
    }
    if (a.accumulate) val += op[c];
    vals[c] = val;
    ssum[c] += val; ssq[c] += val * val;
  }
  if (a.out_vec4) *reinterpret_cast<float4*>(op) = make_float4(vals[0], vals[1], vals[2], vals[3]);
  else {
#pragma unroll
    for (int c = 0; c < NO; ++c) op[c] = vals[c];
  }
}

// per-block statistics row from the lanes' running sums (red: 32 floats of LDS)
template <int NO>
__device__ __forceinline__ void direct_stats(const DArgs& a, int n, const float* ssum, const float* ssq, float* red) {
  if (a.stats == nullptr) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < NO; ++c) {
    const float s = wave_sum(ssum[c]), q = wave_sum(ssq[c]);
    if (lane == 0) { red[(0 * 4 + wave) * 4 + c] = s; red[(1 * 4 + wave) * 4 + c] = q; }
  }
  __syncthreads();
  if (threadIdx.x < NO) {
    const int c = threadIdx.x;
    const float s = red[0 * 4 + c] + red[1 * 4 + c] + red[2 * 4 + c] + red[3 * 4 + c];
    const float q = red[16 + 0 * 4 + c] + red[16 + 1 * 4 + c] + red[16 + 2 * 4 + c] + red[16 + 3 * 4 + c];
    const long long rrow = (long long)n * a.blocks_per_n + blockIdx.x;
    a.stats[(rrow * 2 + 0) * a.N + c] = s;
    a.stats[(rrow * 2 + 1) * a.N + c] = q;
  }
}

// ------------------------------------------------------------------ lanes along K (K = 32 or 64)
// The thread-per-voxel form above reads a 16-byte piece of 64 different cache lines with every load once K is
// wide (the full-resolution ConvTranspose3d 64->R of the U-Net, the 1x1 final conv 32->R of the deep-fusion
// net).  Here KL = K/4 adjacent lanes own one voxel's channels (one 16-byte load each: a voxel is one contiguous
// K*4-byte run), a wave walks 64 voxels of one output row in KL passes, and the taps are the OUTER loop so a
// tap's 4x4 weights are read from LDS once per wave, not once per voxel.  Stride-2 transposed taps depend on the
// output parity, so a wave takes voxels of one x parity: its tap list is wave-uniform.  Addresses are a
// wave-uniform row pointer plus a 32-bit lane offset that advances by a scalar per pass; rows whose 64 voxels
// are all in range (all but the image border) take a path without per-voxel predicates.  Partial sums
// (KL lanes x NO outputs x KL passes) are combined by a halving exchange: log2(KL) steps, each lane ends with
// the NO outputs of one voxel.  A workgroup keeps its weights in LDS and walks a contiguous range of units;
// ranges that are neighbours in the volume run on the same XCD (xcd_contiguous_id) so halo rows hit its L2.
// XBF: the input is bf16-stored (the 1x1x1 head of the deep-fusion net reads a wide forward activation)
template <int KL, int NO, bool HAS_T, bool XBF = false>
__global__ __launch_bounds__(256) void direct_klane_kernel(DArgs a) {
  extern __shared__ float lds[];
  constexpr int VW = 64 / KL;            // voxels per pass
  constexpr int NP = KL;                 // passes: 64 voxels per wave
  const int T = a.ksize * a.ksize * a.ksize;
  float* wl = lds;                       // [T][K][4]
  float* red = lds + T * a.K * 4;        // [sum|sq][wave][channel]
  const int n = blockIdx.y;
  a.w = pset_packed(a.ps, a.w, n); a.bias = pset_bias(a.ps, a.bias, n);      // this batch item's parameter set
  for (int i = threadIdx.x; i < T * a.K; i += 256)
    *reinterpret_cast<float4*>(wl + 4 * i) = *reinterpret_cast<const float4*>(a.w + 4 * i);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int kl = lane % KL, g = lane / KL, k0 = kl * 4;
  float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
  if (HAS_T) {
    nl_coeff_vec<4>(a.tin, n, a.K, k0, sc, sh);
  }
  const int pad = (a.ksize - 1) / 2;
  const bool s2 = a.stride == 2;
  const bool s2t = a.transposed && s2;
  const int xstep = s2t ? 2 : 1;
  const int rowlen = (a.out.w + xstep - 1) / xstep;
  const int chunks = (rowlen + 63) / 64;
  const long long units = (long long)a.out.d * a.out.h * xstep * chunks;
  constexpr unsigned EB = XBF ? 2u : 4u;              // bytes per input element
  const unsigned sw4 = (unsigned)a.in.sw * EB;        // bytes between x neighbours of the input
  auto ld4 = [](const char* p) {                       // 4 consecutive input channels
    if constexpr (XBF) {
      const uint2 u = *reinterpret_cast<const uint2*>(p);
      return make_float4(bf16_bits_to_f32(u.x & 0xffffu), bf16_bits_to_f32(u.x >> 16), bf16_bits_to_f32(u.y & 0xffffu), bf16_bits_to_f32(u.y >> 16));
    } else {
      return *reinterpret_cast<const float4*>(p);
    }
  };
  float ssum[NO], ssq[NO];
#pragma unroll
  for (int c = 0; c < NO; ++c) { ssum[c] = 0.f; ssq[c] = 0.f; }

  long long ufirst, ulast;
  unit_range(units, xcd_contiguous_id(blockIdx.x, gridDim.x), gridDim.x, ufirst, ulast);
  for (long long u0 = ufirst + wave; u0 < ulast; u0 += 4) {
    long long u = u0;
    const int chunk = (int)(u % chunks); u /= chunks;
    const int px = (int)(u % xstep); u /= xstep;
    const int oy = (int)(u % a.out.h);
    const int oz = (int)(u / a.out.h);
    const int ibase = chunk * 64;                        // first voxel index (within the parity row) of this wave
    float acc[NP * NO];
#pragma unroll
    for (int i = 0; i < NP * NO; ++i) acc[i] = 0.f;
    const char* inb = reinterpret_cast<const char*>(a.in.p) + (long long)n * a.in.sn * EB;
    for (int kz = 0; kz < a.ksize; ++kz) {
      int iz;
      if (!a.transposed) iz = oz * a.stride + kz - pad;
      else {
        const int t = oz + pad - kz;
        if (t < 0 || (s2 && (t & 1))) continue;
        iz = s2 ? (t >> 1) : t;
      }
      if ((unsigned)iz >= (unsigned)a.in.d) continue;
      for (int ky = 0; ky < a.ksize; ++ky) {
        int iy;
        if (!a.transposed) iy = oy * a.stride + ky - pad;
        else {
          const int t = oy + pad - ky;
          if (t < 0 || (s2 && (t & 1))) continue;
          iy = s2 ? (t >> 1) : t;
        }
        if ((unsigned)iy >= (unsigned)a.in.h) continue;
        const char* row = inb + ((long long)iz * a.in.sd + (long long)iy * a.in.sh) * EB;
        for (int kx = 0; kx < a.ksize; ++kx) {
          // input x of voxel i of this wave's row: ix = xa*i + xb (wave-uniform xa, xb)
          int xa, xb;
          if (!a.transposed) { xa = a.stride; xb = kx - pad; }
          else if (s2) { if ((px + pad - kx) & 1) continue; xa = 1; xb = (px + pad - kx) >> 1; }
          else { xa = 1; xb = pad - kx; }
          const float* wt = wl + (((kz * a.ksize + ky) * a.ksize + kx) * a.K + k0) * 4;
          float w[4][4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float4 t4 = *reinterpret_cast<const float4*>(wt + 4 * j);
            w[j][0] = t4.x; w[j][1] = t4.y; w[j][2] = t4.z; w[j][3] = t4.w;
          }
          const int ilast = ibase + 63;
          const bool fast = px + xstep * ilast < a.out.w && xa * ibase + xb >= 0 && xa * ilast + xb < a.in.w;
          if (fast) {
            unsigned boff = (unsigned)(xa * (ibase + g) + xb) * sw4 + (unsigned)k0 * EB;
            const unsigned bstep = (unsigned)(xa * VW) * sw4;
#pragma unroll
            for (int p = 0; p < NP; ++p) {
              const float4 x4 = ld4(row + boff);
              boff += bstep;
              const float xs[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const float xv = HAS_T ? nl_apply(xs[j], sc[j], sh[j], a.tin.relu) : xs[j];
#pragma unroll
                for (int c = 0; c < NO; ++c) acc[p * NO + c] = fmaf(xv, w[j][c], acc[p * NO + c]);
              }
            }
          } else {
#pragma unroll
            for (int p = 0; p < NP; ++p) {
              const int i = ibase + p * VW + g;
              const int ix = xa * i + xb;
              const bool ok = px + xstep * i < a.out.w && (unsigned)ix < (unsigned)a.in.w;
              const float4 x4 = ld4(row + (unsigned)(ok ? ix : 0) * sw4 + (unsigned)k0 * EB);
              const float xs[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                float xv = HAS_T ? nl_apply(xs[j], sc[j], sh[j], a.tin.relu) : xs[j];
                xv = ok ? xv : 0.f;
#pragma unroll
                for (int c = 0; c < NO; ++c) acc[p * NO + c] = fmaf(xv, w[j][c], acc[p * NO + c]);
              }
            }
          }
        }
      }
    }
    // ---- halving exchange inside each KL-lane group: lane kl ends with pass kl's NO sums
#pragma unroll
    for (int m = KL / 2, h = NP * NO / 2; m >= 1; m >>= 1, h >>= 1) {
      const bool hi = (kl & m) != 0;
#pragma unroll
      for (int i = 0; i < NP * NO / 2; ++i) {
        if (i < h) {
          const float send = hi ? acc[i] : acc[i + h];
          const float keep = hi ? acc[i + h] : acc[i];
          acc[i] = keep + __shfl_xor(send, m, 64);
        }
      }
    }
    // ---- this lane owns voxel (ibase + kl*VW + g) of the wave's row
    const int ox = px + xstep * (ibase + kl * VW + g);
    if (ox < a.out.w) direct_epilogue<NO>(a, n, oz, oy, ox, acc, ssum, ssq);
  }
  direct_stats<NO>(a, n, ssum, ssq, red);
}

// ------------------------------------------------------------------ thread per voxel, K <= 4, k3 s1
// The full-resolution R->R convolutions of the top ResidualUnit (forward and input gradient): a voxel's K
// channels are ONE 16-byte load (channel rows are padded to 4 floats); weights are wave-uniform and come through
// the scalar cache (constant address space: s_load, SGPR operands of the FMAs).  A wave computes 64 consecutive x
// of TWO adjacent output rows: the four input rows of a z-plane are loaded once for both (12 independent 16-byte
// loads per plane, issued before any use) and every weight is fetched once per pair.  Rows away from the z / y
// border (97 % at 128^3) take a path without any row test; the x border is resolved once per voxel by keeping one
// partial sum per kx and dropping the invalid ones at the end.  The input gradient is the same loop with the tap
// index mirrored (26 - tap).
typedef const __attribute__((address_space(4))) float cfloat;
typedef float cf16 __attribute__((ext_vector_type(16)));
typedef float cf4 __attribute__((ext_vector_type(4)));

template <int KI, int NO, bool HAS_T>
__global__ __launch_bounds__(256) void direct_row_kernel(DArgs a) {
  __shared__ float red[32];
  const int n = blockIdx.y;
  a.w = pset_packed(a.ps, a.w, n); a.bias = pset_bias(a.ps, a.bias, n);      // this batch item's parameter set
  cfloat* wc = (cfloat*)a.w;             // [27][KI][4]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float sc[KI], sh[KI];
#pragma unroll
  for (int k = 0; k < KI; ++k) { sc[k] = 1.f; sh[k] = 0.f; }
  if (HAS_T) nl_coeff_vec<KI>(a.tin, n, KI, 0, sc, sh);
  const int hp = (a.out.h + 1) / 2;
  const int chunks = (a.out.w + 63) / 64;
  const long long units = (long long)a.out.d * hp * chunks;
  const int sw4 = a.in.sw * 4;
  float ssum[NO], ssq[NO];
#pragma unroll
  for (int c = 0; c < NO; ++c) { ssum[c] = 0.f; ssq[c] = 0.f; }
  long long ufirst, ulast;
  unit_range(units, xcd_contiguous_id(blockIdx.x, gridDim.x), gridDim.x, ufirst, ulast);
  const char* inb = reinterpret_cast<const char*>(a.in.p + (long long)n * a.in.sn);
  for (long long u0 = ufirst + wave; u0 < ulast; u0 += 4) {
    long long u = u0;
    const int chunk = (int)(u % chunks); u /= chunks;
    const int oy0 = 2 * (int)(u % hp);
    const int oz = (int)(u / hp);
    const int ox = chunk * 64 + lane;
    // byte offsets of the three x neighbours inside a row, clamped into the row (invalid ones are dropped below)
    unsigned boff[3];
    bool okx[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int ix = ox + dx - 1;
      okx[dx] = (unsigned)ix < (unsigned)a.in.w && ox < a.out.w;
      boff[dx] = (unsigned)(min(max(ix, 0), a.in.w - 1) * sw4);
    }
    float acc[2][3][NO];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int c = 0; c < NO; ++c) acc[j][dx][c] = 0.f;
    const bool interior = oz >= 1 && oz + 1 < a.in.d && oy0 >= 1 && oy0 + 2 < a.in.h;
#pragma unroll
    for (int dz = 0; dz < 3; ++dz) {
      const int iz = oz + dz - 1;
      if (!interior && (unsigned)iz >= (unsigned)a.in.d) continue;
      const char* plane = inb + (long long)iz * a.in.sd * 4;
      float4 xr[4][3];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int iy = min(max(oy0 - 1 + r, 0), a.in.h - 1);
        const char* row = plane + (long long)iy * a.in.sh * 4;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) xr[r][dx] = *reinterpret_cast<const float4*>(row + boff[dx]);
      }
      float xs[4][3][KI];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool rok = interior || (unsigned)(oy0 - 1 + r) < (unsigned)a.in.h;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const float raw[4] = {xr[r][dx].x, xr[r][dx].y, xr[r][dx].z, xr[r][dx].w};
#pragma unroll
          for (int k = 0; k < KI; ++k) {
            const float v = HAS_T ? nl_apply(raw[k], sc[k], sh[k], a.tin.relu) : raw[k];
            xs[r][dx][k] = rok ? v : 0.f;
          }
        }
      }
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const int t = (dz * 3 + dy) * 3 + dx;
          cfloat* wt = wc + (a.transposed ? 26 - t : t) * KI * 4;
#pragma unroll
          for (int k = 0; k < KI; ++k) {
            const cf4 wv = *reinterpret_cast<const __attribute__((address_space(4))) cf4*>(wt + k * 4);   // one s_load_dwordx4
#pragma unroll
            for (int c = 0; c < NO; ++c) {
              acc[0][dx][c] = fmaf(xs[dy][dx][k], wv[c], acc[0][dx][c]);
              acc[1][dx][c] = fmaf(xs[dy + 1][dx][k], wv[c], acc[1][dx][c]);
            }
          }
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float val[NO];
#pragma unroll
      for (int c = 0; c < NO; ++c)
        val[c] = (okx[0] ? acc[j][0][c] : 0.f) + (okx[1] ? acc[j][1][c] : 0.f) + (okx[2] ? acc[j][2][c] : 0.f);
      if (ox < a.out.w && oy0 + j < a.out.h) direct_epilogue<NO>(a, n, oz, oy0 + j, ox, val, ssum, ssq);
    }
  }
  direct_stats<NO>(a, n, ssum, ssq, red);
}

// ------------------------------------------------------------------ the same R -> R convolution on the 4x4x4 matrix tiles
// `bf16` precision mode.  direct_row_kernel spends 27 * KI * NO fp32 FMAs per voxel on the vector ALU (243 for 3 -> 3: at
// 128^3 the two launches of a step cost as much vector-ALU time as a 64^3 32 -> 32 implicit GEMM, for 0.05 % of its MACs)
// and with several volumes in flight vector-ALU issue is what the chip runs out of.  v_mfma_f32_4x4x4_16B_bf16 computes 16
// independent [4 x 4] += [4 x 4][4 x 4] products per instruction: block = 4 voxels (rows), k = the 4 padded input
// channels, columns = the 4 padded output channels - one instruction per tap and 64 voxels, no padding of N to 32.
// Lane maps (checked on the device with integer data): A: lane l holds row l & 3 of block l >> 2 (its 4 k values = ONE
// 8-byte LDS read of a bf16 voxel); B: lane l holds column l & 3 (4 k values, the same for all blocks: 27 taps x 2
// registers per lane, built once from the fp32 image); D: lane l holds rows 0..3 of column l & 3 of block l >> 2.
// Row r of block b is voxel x = 16 r + b of the wave's 64-voxel row, so the store of accumulator register r is 16 voxels
// x 4 channels = 256 contiguous bytes.  The input tile (2 x 8 x 64 voxels + halo) is staged once as bf16 [voxel][4].
// Measured (r02e, 3 -> 3 at 128^3, scripts/experiments/thin_phases.py): 33 us against 43 us for direct_row_kernel (the first
// version, one voxel per staging item and 108 scalar weight loads per lane, took 44: 26 of them index arithmetic), 58.7
// against 58.6 volumes/s with four volumes in flight (+1 % on another box), full-size parity against the fp32 oracle
// unchanged (logits 1.496e-2 vs 1.494e-2 of max): the default of bf16 precision (MMTTA_OPT_THIN_MFMA).
typedef short cs4 __attribute__((ext_vector_type(4)));
typedef float cf4v __attribute__((ext_vector_type(4)));

// GBF: input, output and the fused-add / accumulate operands are bf16-stored with 8-byte voxels (the input-gradient launch
// when the thin gradients are bf16-stored, round 3: the operands are rounded to bf16 while staging either way)
template <bool HAS_T, int TZ, bool GBF = false>
__global__ __launch_bounds__(256) void conv3_mfma4_kernel(DArgs a) {
  // 8 x 8 x 64 tile: the halo box is 1.6x the tile.  The box rows are padded to 68 voxels so that a staging item is a run
  // of 4 voxels along x (one decode, one row address, four 16-byte loads): phase timing of the first version showed 26 of
  // its 44 us in per-voxel index / address arithmetic and in 108 scalar weight loads per lane, not in loads or MFMAs.
  constexpr int TY = 8, TX = 64, BZ = TZ + 2, BY = TY + 2, BX = 68, RUNS = BX / 4, NRUN = BZ * BY * RUNS;
  __shared__ uint2 box[BZ * BY * BX];
  __shared__ uint2 wtab[27 * 4];                // [tap][column] x 4 k values (bf16)
  __shared__ float red[32];
  const int n = blockIdx.y;
  a.w = pset_packed(a.ps, a.w, n); a.bias = pset_bias(a.ps, a.bias, n);      // this batch item's parameter set
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int txn = (a.out.w + TX - 1) / TX, tyn = (a.out.h + TY - 1) / TY;
  int t = blockIdx.x;
  const int txi = t % txn; t /= txn;
  const int tyi = t % tyn;
  const int tzi = t / tyn;
  const int oz0 = tzi * TZ, oy0 = tyi * TY, ox0 = txi * TX;
  // ---- B operand table: column j of tap tp = the 4 k values w[tp][k][j] (fp32 image [tap][K][4], mirrored for the gradient)
  if (tid < 27 * 4) {
    const int tp = tid >> 2, j = tid & 3;
    const float* wt = a.w + (a.transposed ? 26 - tp : tp) * a.K * 4;
    float wk[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) wk[k] = (k < a.K && j < a.N) ? wt[k * 4 + j] : 0.f;
    wtab[tid] = make_uint2(f32x2_to_bf16x2(wk[0], wk[1]), f32x2_to_bf16x2(wk[2], wk[3]));
  }
  {  // ---- stage the halo box: runs of 4 voxels, two runs (8 loads) in flight per thread
    constexpr int NIT = (NRUN + 255) / 256, RND = NIT;      // every load of the box in flight at once (RND = 2: two dependent round trips)
    float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
    if (HAS_T) nl_coeff_vec<4>(a.tin, n, a.K, 0, sc, sh);          // channels >= K: scale = shift = 0
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k >= a.K) { sc[k] = 0.f; sh[k] = 0.f; }                   // pad lanes of the voxel row may hold anything
    const float relu_lo = (HAS_T && a.tin.relu) ? 0.f : -__builtin_inff();
    const float* inb = GBF ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(a.in.p) + (long long)n * a.in.sn)
                           : a.in.p + (long long)n * a.in.sn;
    const unsigned isw = (unsigned)a.in.sw;
#pragma unroll 1
    for (int j0 = 0; j0 < NIT; j0 += RND) {
      float4 raw[RND][4];
      unsigned okm = 0u;
#pragma unroll
      for (int j = 0; j < RND; ++j) {
        const int v = min(tid + 256 * (j0 + j), NRUN - 1);
        const int bz = v / (BY * RUNS), rem = v - bz * (BY * RUNS), by = rem / RUNS, run = rem - by * RUNS;
        const int iz = oz0 - 1 + bz, iy = oy0 - 1 + by, ix = ox0 - 1 + 4 * run;
        const bool rok = (unsigned)iz < (unsigned)a.in.d && (unsigned)iy < (unsigned)a.in.h;
        const long long rowo = (long long)min(max(iz, 0), a.in.d - 1) * a.in.sd + (long long)min(max(iy, 0), a.in.h - 1) * a.in.sh;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          okm |= ((rok && (unsigned)(ix + q) < (unsigned)a.in.w) ? 1u : 0u) << (4 * j + q);
          const long long eo = rowo + (unsigned)min(max(ix + q, 0), a.in.w - 1) * isw;
          if constexpr (GBF) {          // the four bf16 channels travel in two dwords of the raw register
            const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(inb) + eo);
            raw[j][q].x = __uint_as_float(u.x); raw[j][q].y = __uint_as_float(u.y);
          } else {
            raw[j][q] = *reinterpret_cast<const float4*>(inb + eo);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < RND; ++j) {
        uint2 pk[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if constexpr (GBF) {
            const unsigned ux = __float_as_uint(raw[j][q].x), uy = __float_as_uint(raw[j][q].y);
            raw[j][q] = make_float4(bf16_bits_to_f32(ux & 0xffffu), __uint_as_float(ux & 0xffff0000u),
                                    bf16_bits_to_f32(uy & 0xffffu), __uint_as_float(uy & 0xffff0000u));
          }
          const float xs[4] = {raw[j][q].x, raw[j][q].y, raw[j][q].z, raw[j][q].w};
          float v4[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) v4[k] = fmaxf(fmaf(xs[k], sc[k], sh[k]), relu_lo);
          const unsigned m = ((okm >> (4 * j + q)) & 1u) ? 0xffffffffu : 0u;
          pk[q].x = f32x2_to_bf16x2(v4[0], v4[1]) & m; pk[q].y = f32x2_to_bf16x2(v4[2], v4[3]) & m;
        }
        if (tid + 256 * (j0 + j) < NRUN) {
          // run v = voxels 4 v .. 4 v + 3 of its box row.  The first 64 voxels of a row are stored PERMUTED, voxel xi at slot
          // (xi % 16) * 4 + xi / 16: the A operand of lane (blk, arow) is voxel 16 arow + blk (+ tap), and in plain order
          // arow 0 / 2 and 1 / 3 met in the same banks (SQ_LDS_BANK_CONFLICT: 42 % of the launch's cycles per CU);
          // permuted, the 64 lanes of a tap read 64 consecutive slots
          const int v = tid + 256 * (j0 + j), brow = v / RUNS, run = v - brow * RUNS;
          uint2* drow = box + brow * BX;
#pragma unroll
          for (int q = 0; q < 4; ++q) drow[run < 16 ? 16 * (run & 3) + 4 * q + (run >> 2) : 64 + q] = pk[q];
        }
      }
    }
  }
  const int jc = lane & 3, blk = lane >> 2;
  float bias = 0.f, asc = 1.f, ash = 0.f;
  if (jc < a.N) {
    if (a.bias) bias = a.bias[jc];
    if (a.add) nl_coeff(a.tadd, n, a.N, jc, asc, ash);
  }
  __syncthreads();
  cs4 wb[27];
#pragma unroll
  for (int tp = 0; tp < 27; ++tp) wb[tp] = __builtin_bit_cast(cs4, wtab[tp * 4 + jc]);
  float ssum = 0.f, ssq = 0.f;
  const int arow = lane & 3;                    // A: this lane supplies row `arow` of block `blk`: voxel x = 16 arow + blk
  const int jl = min(jc, a.N - 1);
  // element offsets of this lane's four output voxels inside a row (x = ox0 + 16 r + blk), and whether they exist
  unsigned xoff_o[4], xoff_a[4];
  unsigned xok = 0u;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int ox = ox0 + 16 * r + blk;
    xok |= (ox < a.out.w ? 1u : 0u) << r;
    xoff_o[r] = (unsigned)min(ox, a.out.w - 1) * (unsigned)a.out.sw;
    xoff_a[r] = (unsigned)min(ox, a.out.w - 1) * (unsigned)a.asw;
  }
  // the fused-add / accumulate operands of trip rp + 1 are requested before the MFMAs of trip rp: behind the MFMA loop
  // every trip waited a full memory latency for them (both 3 -> 3 launches of a step carry the residual add)
  float addn[2][4], oldn[2][4];
  auto fetch = [&](int rp) {
    const int r0 = rp * 8 + wave * 2;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int oz = min(oz0 + (r0 + q) / TY, a.out.d - 1), oy = min(oy0 + (r0 + q) % TY, a.out.h - 1);
#pragma unroll
      for (int r = 0; r < 4; ++r) { addn[q][r] = 0.f; oldn[q][r] = 0.f; }
      if (a.add) {
        const long long ao = (long long)n * a.asn + (long long)oz * a.asd + (long long)oy * a.ash;
#pragma unroll
        for (int r = 0; r < 4; ++r) addn[q][r] = ld1_t<GBF>(a.add, ao + xoff_a[r] + jl);
      }
      if (a.accumulate) {
        const long long oo = (long long)n * a.out.sn + (long long)oz * a.out.sd + (long long)oy * a.out.sh;
#pragma unroll
        for (int r = 0; r < 4; ++r) oldn[q][r] = ld1_t<GBF>(a.out.p, oo + xoff_o[r] + jl);
      }
    }
  };
  fetch(0);
#pragma unroll 1
  for (int rp = 0; rp < TZ * TY / 8; ++rp) {    // two rows of the tile per wave and trip (independent accumulator chains)
    const int r0 = rp * 8 + wave * 2;
    float addc[2][4], oldc[2][4];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) { addc[q][r] = addn[q][r]; oldc[q][r] = oldn[q][r]; }
    if (rp + 1 < TZ * TY / 8) fetch(rp + 1);
    cf4v acc[2];
    const uint2* ab[2][3];                        // per x tap: the lane's slot of voxel 16 arow + blk + dx in the permuted row
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int zl = (r0 + q) / TY, yl = (r0 + q) % TY;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int xi = 16 * arow + blk + dx;
        ab[q][dx] = box + (zl * BY + yl) * BX + (xi < 64 ? ((xi & 15) << 2) | (xi >> 4) : xi);
      }
      acc[q] = cf4v{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int tp = 0; tp < 27; ++tp) {
      const int toff = ((tp / 9) * BY + (tp / 3) % 3) * BX;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const uint2 av = ab[q][tp % 3][toff];
        acc[q] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(cs4, av), wb[tp], acc[q], 0, 0, 0);
      }
    }
    // ---- epilogue: register r of lane (blk, jc) = output voxel x = 16 r + blk, channel jc
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int oz = oz0 + (r0 + q) / TY, oy = oy0 + (r0 + q) % TY;
      const bool rowok = oz < a.out.d && oy < a.out.h;
      const long long orowo = (long long)n * a.out.sn + (long long)min(oz, a.out.d - 1) * a.out.sd + (long long)min(oy, a.out.h - 1) * a.out.sh;
      float* orow = a.out.p + orowo;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[q][r] + bias;
        if (a.add) v += nl_apply(addc[q][r], asc, ash, a.tadd.relu);
        if (a.accumulate) v += oldc[q][r];
        if constexpr (GBF) {
          // 8-byte voxels: channel lanes pair up, the even one stores both as one dword (a 2-byte store per lane is a partial
          // dword write); channels >= N are stored as zeros (the rows own their pad)
          v = jc < a.N ? v : 0.f;
          const float other = __shfl_xor(v, 1, 64);
          if (rowok && ((xok >> r) & 1u)) {
            if (!(jc & 1))
              *reinterpret_cast<unsigned int*>(reinterpret_cast<unsigned short*>(a.out.p) + orowo + xoff_o[r] + jc) = f32x2_to_bf16x2(v, other);
            if (jc < a.N) { ssum += v; ssq += v * v; }
          }
          continue;
        }
        if (rowok && ((xok >> r) & 1u)) {
          if (jc < a.N) {
            orow[xoff_o[r] + jc] = v;
            ssum += v; ssq += v * v;
          } else if (a.out_vec4) {
            orow[xoff_o[r] + jc] = 0.f;          // the view owns its pad lanes: keep them zero
          }
        }
      }
    }
  }
  if (a.stats != nullptr) {   // lanes with the same l & 3 hold the same channel
#pragma unroll
    for (int o = 4; o < 64; o <<= 1) { ssum += __shfl_xor(ssum, o, 64); ssq += __shfl_xor(ssq, o, 64); }
    if (lane < 4) { red[(0 * 4 + wave) * 4 + lane] = ssum; red[(1 * 4 + wave) * 4 + lane] = ssq; }
    __syncthreads();
    if (tid < a.N) {
      const float s2 = red[0 * 4 + tid] + red[1 * 4 + tid] + red[2 * 4 + tid] + red[3 * 4 + tid];
      const float q2 = red[16 + 0 * 4 + tid] + red[16 + 1 * 4 + tid] + red[16 + 2 * 4 + tid] + red[16 + 3 * 4 + tid];
      const long long rrow = (long long)n * a.blocks_per_n + blockIdx.x;
      a.stats[(rrow * 2 + 0) * a.N + tid] = s2;
      a.stats[(rrow * 2 + 1) * a.N + tid] = q2;
    }
  }
}

// ------------------------------------------------------------------ stride-2 up-convolution to <= 4 channels
// ConvTranspose3d K -> R (k3, s2, p1, output_padding 1) with K = 32 / 64: the full-resolution last layer of the U-Net.
// out[2i+p] needs in[i] (p = 0: tap 1; p = 1: tap 2) and, for p = 1, in[i+1] (tap 0), per axis.  A workgroup stages
// the 2 x 2 input rows (iz, iz+1) x (iy, iy+1), 65 voxels each, ONCE in LDS (norm+ReLU applied, borders written
// as zeros) and produces the 2 x 2 output rows x 128 voxels that depend on nothing else.  A wave takes one of the
// 8 output parity classes at a time (64 voxels of that class, one per lane): its tap list is wave-uniform, so
// the weights are SGPR operands (constant address space, 64 bytes per load) and every lane simply walks the K
// channels of its 1 / 2 / 4 / 8 input voxels in LDS (16-byte reads, conflict-free at a 16-byte row pad).  No
// cross-lane reduction, each input voxel is fetched from L2 once per workgroup instead of once per tap.
template <int K, int NO, bool HAS_T>
__global__ __launch_bounds__(256) void direct_upconv_kernel(DArgs a) {
  extern __shared__ float lds[];
  constexpr int VS = K + 4;                    // LDS voxel stride (floats)
  constexpr int XV = 65;                       // staged voxels per row
  float* red = lds + 4 * XV * VS;
  const int n = blockIdx.y;
  a.w = pset_packed(a.ps, a.w, n); a.bias = pset_bias(a.ps, a.bias, n);      // this batch item's parameter set
  cfloat* wc = (cfloat*)a.w;                   // [27][K][4]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int chunks = (a.in.w + 63) / 64;
  int b = blockIdx.x;
  const int chunk = b % chunks; b /= chunks;
  const int iy0 = b % a.in.h;
  const int iz0 = b / a.in.h;
  const int ix0 = chunk * 64;
  {  // ---- stage: thread = (channel group of 4, voxel slots); all loads first (clamped), then transform + store
    constexpr int CG = K / 4;                  // channel groups per voxel (256 % CG == 0)
    constexpr int NIT = (4 * XV * CG + 255) / 256;
    const int cg = tid % CG;
    float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
    if (HAS_T) nl_coeff_vec<4>(a.tin, n, K, cg * 4, sc, sh);
    const float* inb = a.in.p + (long long)n * a.in.sn + cg * 4;
    float4 raw[NIT];
#pragma unroll
    for (int q = 0; q < NIT; ++q) {
      const int vs = min(tid / CG + q * (256 / CG), 4 * XV - 1);     // voxel slot: row * XV + x
      const int row = vs / XV, xl = vs % XV;
      const int iz = min(iz0 + (row >> 1), a.in.d - 1), iy = min(iy0 + (row & 1), a.in.h - 1), ix = min(ix0 + xl, a.in.w - 1);
      raw[q] = *reinterpret_cast<const float4*>(inb + (long long)iz * a.in.sd + (long long)iy * a.in.sh + (long long)ix * a.in.sw);
    }
#pragma unroll
    for (int q = 0; q < NIT; ++q) {
      const int vs = tid / CG + q * (256 / CG);
      if (vs < 4 * XV) {
        const int row = vs / XV, xl = vs % XV;
        const bool ok = iz0 + (row >> 1) < a.in.d && iy0 + (row & 1) < a.in.h && ix0 + xl < a.in.w;
        float4 v;
        v.x = ok ? (HAS_T ? nl_apply(raw[q].x, sc[0], sh[0], a.tin.relu) : raw[q].x) : 0.f;
        v.y = ok ? (HAS_T ? nl_apply(raw[q].y, sc[1], sh[1], a.tin.relu) : raw[q].y) : 0.f;
        v.z = ok ? (HAS_T ? nl_apply(raw[q].z, sc[2], sh[2], a.tin.relu) : raw[q].z) : 0.f;
        v.w = ok ? (HAS_T ? nl_apply(raw[q].w, sc[3], sh[3], a.tin.relu) : raw[q].w) : 0.f;
        *reinterpret_cast<float4*>(lds + vs * VS + cg * 4) = v;
      }
    }
  }
  __syncthreads();
  float ssum[NO], ssq[NO];
#pragma unroll
  for (int c = 0; c < NO; ++c) { ssum[c] = 0.f; ssq[c] = 0.f; }
#pragma unroll 1
  for (int rep = 0; rep < 2; ++rep) {
    const int cls = wave + 4 * rep;            // (pz, py, px)
    const int pz = cls >> 2, py = (cls >> 1) & 1, px = cls & 1;
    float acc[NO];
#pragma unroll
    for (int c = 0; c < NO; ++c) acc[c] = 0.f;
    // per axis: parity 0 -> (tap 1, shift 0); parity 1 -> (tap 2, shift 0) and (tap 0, shift 1)
    for (int tz = 0; tz <= pz; ++tz) {
      const int kz = pz == 0 ? 1 : (tz == 0 ? 2 : 0);
      for (int ty = 0; ty <= py; ++ty) {
        const int ky = py == 0 ? 1 : (ty == 0 ? 2 : 0);
        for (int tx = 0; tx <= px; ++tx) {
          const int kx = px == 0 ? 1 : (tx == 0 ? 2 : 0);
          const float* xv = lds + ((tz * 2 + ty) * XV + lane + tx) * VS;
          cfloat* wt = wc + ((kz * 3 + ky) * 3 + kx) * K * 4;
#pragma unroll
          for (int k4 = 0; k4 < K / 4; ++k4) {
            const float4 x4 = *reinterpret_cast<const float4*>(xv + 4 * k4);
            const float xs[4] = {x4.x, x4.y, x4.z, x4.w};
            // the 4 x 4 weights of this channel group as ONE 64-byte scalar load (element-wise indexing makes the
            // compiler fetch the NO used dwords of every row separately: 8x the scalar-memory instructions)
            const cf16 wv = *reinterpret_cast<const __attribute__((address_space(4))) cf16*>(wt + 16 * k4);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
              for (int c = 0; c < NO; ++c) acc[c] = fmaf(xs[j], wv[j * 4 + c], acc[c]);
          }
        }
      }
    }
    const int oz = 2 * iz0 + pz, oy = 2 * iy0 + py, ox = 2 * (ix0 + lane) + px;
    if (ix0 + lane < a.in.w && oz < a.out.d && oy < a.out.h && ox < a.out.w)
      direct_epilogue<NO>(a, n, oz, oy, ox, acc, ssum, ssq);
  }
  direct_stats<NO>(a, n, ssum, ssq, red);
}

static void direct_dims(const mmtta_conv_desc* d, int& K, int& N) {
  switch (d->op) {
    case MMTTA_CONV_FWD: case MMTTA_CONVT_FWD: N = d->cout; K = d->cin; break;
    default: N = d->cin; K = d->cout; break;
  }
}

bool direct_applicable(const mmtta_conv_desc* d) {
  int K, N;
  direct_dims(d, K, N);
  const int T = d->ksize * d->ksize * d->ksize;
  // weights + coefficients must fit in LDS next to nothing else: T*K*16 + K*8 bytes
  return N <= 4 && (size_t)T * K * 16 + (size_t)K * 8 + 128 <= 96 * 1024;
}

// The thin-K matrix-core convolution (chan_mfma_kernel) multiplies 7 k-steps of 16 = (4 taps x 4 padded channels) by 32-column
// blocks; its B fragments are part of the PACKED image since round 3 - entry ((s2 * NB + nb) * 64 + lane) = the lane's 8 k
// values of column nb * 32 + (lane & 31), bf16 - written once per optimizer step by the pack kernels.  Every workgroup
// rebuilt them before (56 - 112 scalar loads per entry and thread: a quarter of the kernel's vector instructions).
long long chan_frag_bytes(const mmtta_conv_desc* d) {
  if (!(d->op == MMTTA_CONV_FWD || d->op == MMTTA_CONVT_DGRAD) || d->ksize != 3 || d->dtype != MMTTA_BF16) return 0;
  const int K = d->op == MMTTA_CONV_FWD ? d->cin : d->cout, N = d->op == MMTTA_CONV_FWD ? d->cout : d->cin;
  if (K > 4 || !(N == 32 || N == 64)) return 0;
  return 7LL * (N / 32) * 64 * 16;
}

long long upconv8_image_bytes(const mmtta_conv_desc* d) {
  const bool ok = d->op == MMTTA_CONVT_FWD && d->ksize == 3 && d->stride == 2 && (d->cin == 32 || d->cin == 64) && d->cout <= 4 &&
                  d->dtype == MMTTA_BF16;
  return ok ? (long long)8 * (d->cin / 16) * 64 * 16 : 0;
}

static bool aligned16(const mmtta_tensor* x) {
  return x->sc == 1 && x->sw % 4 == 0 && x->sh % 4 == 0 && x->sd % 4 == 0 && x->sn % 4 == 0 && ((uintptr_t)x->ptr) % 16 == 0;
}

// ------------------------------------------------------------------ lanes along N (K <= 4 gathered, N = 32 / 64)
// The first layers (Conv3d 4->32 k3 s2 and its strided residual conv) and the input gradient of the last
// up-convolution (ConvTranspose3d 64->R: a stride-2 gather of R = 3 channels into 64).  Their whole reduction is
// 27 taps x <= 4 channels, so a lane owns ONE output channel and keeps its 27 x KI weights in registers; the
// input halo box of a 4x4x8 output tile (16-byte voxels) sits in LDS and is read by broadcast (every lane of a
// voxel's group reads the same 16 bytes); one coalesced store per voxel; the statistics of the following norm are
// plain per-lane sums.  On the fp32 matrix cores these layers ran with 4 of 8 staged channels empty.
struct CArgs {
  TV in; NL tin;
  TV out;
  const float* w; int Kp, Np;     // implicit-GEMM fp32 image [27][Kp][Np]
  const float* bias;
  const float* add; long long asn, asd, ash, asw; NL tadd; int add_bf;
  int accumulate;
  float* stats; int tiles_per_n;
  int tz, ty, tx;
  PSets ps;                       // per-item parameter sets: w / bias are set 0's (common.h)
  int koff;                       // KI == 1: the channel sits at float `koff` of the 16-byte group `in.p` points at
};

template <int S, int KI, int NLN, bool HAS_T>
__global__ __launch_bounds__(256) void direct_chan_kernel(CArgs a) {
  constexpr int TZ = 4, TY = 4, TX = 8;
  constexpr int BZ = (TZ - 1) * S + 3, BY = (TY - 1) * S + 3, BX = (TX - 1) * S + 3, BOX = BZ * BY * BX;
  constexpr int VP = 64 / NLN, NPASS = 32 / VP;          // voxels per pass, passes per wave (32 voxels per wave)
  __shared__ float4 box[BOX];
  __shared__ float red[2][4][NLN];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int t = blockIdx.x;
  const int tile_in_n = t % a.tiles_per_n;
  const int txi = t % a.tx; t /= a.tx;
  const int tyi = t % a.ty; t /= a.ty;
  const int tzi = t % a.tz;
  const int n = t / a.tz;
  a.w = pset_packed(a.ps, a.w, n); a.bias = pset_bias(a.ps, a.bias, n);      // this batch item's parameter set
  const int oz0 = tzi * TZ, oy0 = tyi * TY, ox0 = txi * TX;
  const int iz0 = oz0 * S - 1, iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;
  {  // ---- stage the halo box: all loads (clamped) first, then transform / zero-fill.  A thread keeps one box column and
     // walks box rows (see chan_mfma_kernel): one x clamp, 32-bit offsets, validity as a bit mask.
    constexpr int RPP = 256 / BX, NROW = BZ * BY, NQ = (NROW + RPP - 1) / RPP;
    float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
    if (HAS_T) nl_coeff_vec<4>(a.tin, n, KI, 0, sc, sh);
    const float* inb = a.in.p + (long long)n * a.in.sn;
    const int bx = tid % BX, r0 = tid / BX;
    const int ix = ix0 + bx;
    const bool xok = (unsigned)ix < (unsigned)a.in.w && r0 < RPP;
    const unsigned xoff = (unsigned)min(max(ix, 0), a.in.w - 1) * (unsigned)a.in.sw;
    const unsigned sd = (unsigned)a.in.sd, shh = (unsigned)a.in.sh;
    float4 raw[NQ];
    unsigned okm = 0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int row = min(r0 + q * RPP, NROW - 1);
      const int by = row % BY, bz = row / BY;
      const int iz = iz0 + bz, iy = iy0 + by;
      const bool ok = xok && (unsigned)iz < (unsigned)a.in.d && (unsigned)iy < (unsigned)a.in.h;
      okm |= (ok ? 1u : 0u) << q;
      raw[q] = *reinterpret_cast<const float4*>(inb + ((unsigned)min(max(iz, 0), a.in.d - 1) * sd +
                                                       (unsigned)min(max(iy, 0), a.in.h - 1) * shh + xoff));
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int row = r0 + q * RPP;
      if (r0 < RPP && row < NROW) {
        const bool ok = (okm >> q) & 1u;
        float r4[4] = {raw[q].x, raw[q].y, raw[q].z, raw[q].w};
        if (KI == 1) r4[0] = a.koff == 0 ? raw[q].x : a.koff == 1 ? raw[q].y : a.koff == 2 ? raw[q].z : raw[q].w;
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
          v[k] = (ok && k < KI) ? (HAS_T ? nl_apply(r4[k], sc[k], sh[k], a.tin.relu) : r4[k]) : 0.f;
        box[row * BX + bx] = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
  }
  // ---- this lane's output channel and its weights
  const int nl = lane % NLN, vslot = lane / NLN;
  const int N = a.out.c;
  const bool nok = nl < N;
  float w[27][KI];
  {
    const float* wp = a.w + min(nl, a.Np - 1);
#pragma unroll
    for (int tp = 0; tp < 27; ++tp)
#pragma unroll
      for (int k = 0; k < KI; ++k) w[tp][k] = wp[((long long)tp * a.Kp + k) * a.Np];
  }
  float bias = 0.f, asc = 1.f, ash = 0.f;
  if (a.bias) bias = a.bias[min(nl, N - 1)];
  if (a.add) nl_coeff(a.tadd, n, N, min(nl, N - 1), asc, ash);
  __syncthreads();
  float ssum = 0.f, ssq = 0.f;
  // a wave owns the z-slice oz0 + wave (32 voxels = TY x TX): 32-bit element offsets inside it (host-checked)
  const int oz = oz0 + wave;
  float* const outp = a.out.p + (long long)n * a.out.sn + (long long)oz * a.out.sd + nl;
  const float* const addp = a.add ? a.add + (long long)n * a.asn + (long long)oz * a.asd + nl : nullptr;
#pragma unroll 2
  for (int p = 0; p < NPASS; ++p) {
    const int q = p * VP + vslot;
    const int xl = q % TX, yl = q / TX, zl = wave;
    const float4* bp = box + ((zl * S) * BY + yl * S) * BX + xl * S;
    float acc = 0.f;
#pragma unroll
    for (int dz = 0; dz < 3; ++dz)
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const float4 x4 = bp[(dz * BY + dy) * BX + dx];
          const float xs[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
          for (int k = 0; k < KI; ++k) acc = fmaf(xs[k], w[(dz * 3 + dy) * 3 + dx][k], acc);
        }
    const int oy = oy0 + yl, ox = ox0 + xl;
    if (nok && oz < a.out.d && oy < a.out.h && ox < a.out.w) {
      float val = acc + bias;
      if (a.add) val += nl_apply(addp[(unsigned)oy * (unsigned)a.ash + (unsigned)ox * (unsigned)a.asw], asc, ash, a.tadd.relu);
      float* op = outp + ((unsigned)oy * (unsigned)a.out.sh + (unsigned)ox * (unsigned)a.out.sw);
      if (a.accumulate) val += *op;
      *op = val;
      ssum += val; ssq += val * val;
    }
  }
  if (a.stats != nullptr) {
    if (NLN == 32) { ssum += __shfl_xor(ssum, 32, 64); ssq += __shfl_xor(ssq, 32, 64); }
    if (lane < NLN) { red[0][wave][lane] = ssum; red[1][wave][lane] = ssq; }
    __syncthreads();
    if (tid < NLN && tid < N) {
      const float s0 = (red[0][0][tid] + red[0][1][tid]) + (red[0][2][tid] + red[0][3][tid]);
      const float s1 = (red[1][0][tid] + red[1][1][tid]) + (red[1][2][tid] + red[1][3][tid]);
      const long long row = (long long)n * a.tiles_per_n + tile_in_n;
      a.stats[(row * 2 + 0) * N + tid] = s0;
      a.stats[(row * 2 + 1) * N + tid] = s1;
    }
  }
}

// The same layers on the matrix cores (`bf16` precision mode): the 27 taps x 4 (padded) channels are the reduction axis
// of v_mfma_f32_32x32x16_bf16, 7 k-steps of 16 = four taps each (tap 27 is a zero pad).  A = 32 voxels of the tile (one
// z-slice per wave) gathered from the bf16 halo box, two 8-byte LDS reads (two taps) per lane and k-step; B = the lane's
// output channel, 7 x NB fragments kept in registers for the whole tile; the accumulator layout (column = channel = lane
// & 31) is already "lanes along N", so stores stay 128-byte rows and the statistics per-lane sums.
// XBF: the gathered <= 4-channel tensor is bf16-stored, 8 bytes a voxel (the network input of bf16 precision, round 3: this
// kernel and the thin weight gradient round it to bf16 while staging anyway - the same values from half the bytes)
template <int S, int KI, int NB, bool HAS_T, bool XBF = false>
__global__ __launch_bounds__(256) void chan_mfma_kernel(CArgs a) {
  constexpr int TZ = 4, TY = 4, TX = 8;
  constexpr int BZ = (TZ - 1) * S + 3, BY = (TY - 1) * S + 3, BX = (TX - 1) * S + 3, BOX = BZ * BY * BX;
  __shared__ uint2 box[BOX];                       // 4 bf16 channels per voxel
  __shared__ float red[2][4][32 * NB];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, r = lane & 31;
  int t = blockIdx.x;
  const int tile_in_n = t % a.tiles_per_n;
  const int txi = t % a.tx; t /= a.tx;
  const int tyi = t % a.ty; t /= a.ty;
  const int tzi = t % a.tz;
  const int n = t / a.tz;
  a.w = pset_packed(a.ps, a.w, n); a.bias = pset_bias(a.ps, a.bias, n);      // this batch item's parameter set
  const int oz0 = tzi * TZ, oy0 = tyi * TY, ox0 = txi * TX;
  const int iz0 = oz0 * S - 1, iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;
  const int N = a.out.c;
  {  // ---- stage the halo box: all loads (clamped) first, then transform / zero-fill / round to bf16.
     // A thread keeps one box column bx and walks box rows (bz, by): the x clamp and mask are computed once, a row costs one
     // small decode, offsets are 32-bit elements from the batch item (host-checked), validity travels as a bit mask.
    constexpr int RPP = 256 / BX, NROW = BZ * BY, NQ = (NROW + RPP - 1) / RPP;
    float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
    if (HAS_T) nl_coeff_vec<4>(a.tin, n, KI, 0, sc, sh);
    const float* inb = XBF ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(a.in.p) + (long long)n * a.in.sn)
                           : a.in.p + (long long)n * a.in.sn;
    const int bx = tid % BX, r0 = tid / BX;
    const int ix = ix0 + bx;
    const bool xok = (unsigned)ix < (unsigned)a.in.w && r0 < RPP;
    const unsigned xoff = (unsigned)min(max(ix, 0), a.in.w - 1) * (unsigned)a.in.sw;
    const unsigned sd = (unsigned)a.in.sd, shh = (unsigned)a.in.sh;
    float4 raw[NQ];
    uint2 rawh[NQ];
    unsigned okm = 0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int row = min(r0 + q * RPP, NROW - 1);
      const int by = row % BY, bz = row / BY;
      const int iz = iz0 + bz, iy = iy0 + by;
      const bool ok = xok && (unsigned)iz < (unsigned)a.in.d && (unsigned)iy < (unsigned)a.in.h;
      okm |= (ok ? 1u : 0u) << q;
      const unsigned off = (unsigned)min(max(iz, 0), a.in.d - 1) * sd + (unsigned)min(max(iy, 0), a.in.h - 1) * shh + xoff;
      if constexpr (XBF) rawh[q] = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(inb) + off);
      else raw[q] = *reinterpret_cast<const float4*>(inb + off);
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int row = r0 + q * RPP;
      if (r0 < RPP && row < NROW) {
        const bool ok = (okm >> q) & 1u;
        if constexpr (XBF && !HAS_T && KI == 4) {           // already what the image holds: mask and store
          const unsigned m = ok ? 0xffffffffu : 0u;
          box[row * BX + bx] = make_uint2(rawh[q].x & m, rawh[q].y & m);
          continue;
        }
        if constexpr (XBF)
          raw[q] = make_float4(bf16_bits_to_f32(rawh[q].x & 0xffffu), __uint_as_float(rawh[q].x & 0xffff0000u),
                               bf16_bits_to_f32(rawh[q].y & 0xffffu), __uint_as_float(rawh[q].y & 0xffff0000u));
        float r4[4] = {raw[q].x, raw[q].y, raw[q].z, raw[q].w};
        if (KI == 1) r4[0] = a.koff == 0 ? raw[q].x : a.koff == 1 ? raw[q].y : a.koff == 2 ? raw[q].z : raw[q].w;
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
          v[k] = (ok && k < KI) ? (HAS_T ? nl_apply(r4[k], sc[k], sh[k], a.tin.relu) : r4[k]) : 0.f;
        uint2 pk;
        pk.x = __builtin_bit_cast(unsigned int, __builtin_convertvector((float2_t){v[0], v[1]}, bf16x2_t));
        pk.y = __builtin_bit_cast(unsigned int, __builtin_convertvector((float2_t){v[2], v[3]}, bf16x2_t));
        box[row * BX + bx] = pk;
      }
    }
  }
  // ---- B fragments: k = s*16 + h*8 + e  <->  tap = s*4 + h*2 + (e >> 2), channel = e & 3; column = nb*32 + r: the
  // fragment-ordered bf16 image behind the fp32 tap image of this batch item's packed weights (chan_frag_bytes, written by
  // the pack kernels): one 16-byte load per fragment and lane, issued before the box is staged
  uint4 wfrag[7][NB];
  {
    const uint4* fr = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(a.w) + (size_t)27 * a.Kp * a.Np * 4) + lane;
#pragma unroll
    for (int s2 = 0; s2 < 7; ++s2)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) wfrag[s2][nb] = fr[(s2 * NB + nb) * 64];
  }
  float bias[NB], asc[NB], ash[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int col = min(nb * 32 + r, N - 1);
    bias[nb] = a.bias ? a.bias[col] : 0.f;
    asc[nb] = 1.f; ash[nb] = 0.f;
    if (a.add) nl_coeff(a.tadd, n, N, col, asc[nb], ash[nb]);
  }
  __syncthreads();
  // ---- A fragments: MFMA row m = r  <->  voxel (zl = wave, yl = m / 8, xl = m % 8)
  ufloat16 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;
  {
    const int xl = r % TX, yl = r / TX;
    const uint2* bp = box + ((wave * S) * BY + yl * S) * BX + xl * S;
    uint2 av[7][2];
#pragma unroll
    for (int s2 = 0; s2 < 7; ++s2)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int tap = min(s2 * 4 + h * 2 + j, 26);           // tap 27: any valid address, its weights are zero
        const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
        av[s2][j] = bp[(dz * BY + dy) * BX + dx];
      }
#pragma unroll
    for (int s2 = 0; s2 < 7; ++s2) {
      const uint4 a4 = make_uint4(av[s2][0].x, av[s2][0].y, av[s2][1].x, av[s2][1].y);
      const ubf16x8 af = __builtin_bit_cast(ubf16x8, a4);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(ubf16x8, wfrag[s2][nb]), acc[nb], 0, 0, 0);
    }
  }
  // ---- epilogue: accumulator i of lane (h, r) = voxel m = (i & 3) + 8 * (i >> 2) + 4 * h, channel nb*32 + r
  float ssum[NB], ssq[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) { ssum[nb] = 0.f; ssq[nb] = 0.f; }
  // voxel m of accumulator i: row oy0 + (i >> 2), column ox0 + (i & 3) + 4 * h; the z-slice (wave) is uniform.  Offsets are
  // 32-bit elements from the slice (host-checked): four row terms plus four column terms, one add per store.
  const int oz = oz0 + wave;
  const bool zok = oz < a.out.d;
  unsigned oyo[4], oxo[4], ayo[4], axo[4];
  unsigned okrow = 0, okcol = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int oy = oy0 + j, ox = ox0 + j + 4 * h;
    okrow |= (zok && oy < a.out.h ? 1u : 0u) << j;
    okcol |= (ox < a.out.w ? 1u : 0u) << j;
    oyo[j] = (unsigned)oy * (unsigned)a.out.sh; oxo[j] = (unsigned)ox * (unsigned)a.out.sw;
    ayo[j] = (unsigned)oy * (unsigned)a.ash; axo[j] = (unsigned)ox * (unsigned)a.asw;
  }
  const long long obase = (long long)n * a.out.sn + (long long)oz * a.out.sd;
  const long long abase = (long long)n * a.asn + (long long)oz * a.asd;
  MMTTA_BF_DISPATCH(a.out.bf, OBF, {
    float* const outp = OBF ? reinterpret_cast<float*>(reinterpret_cast<unsigned short*>(a.out.p) + obase) : a.out.p + obase;
    const float* const addp = a.add == nullptr ? nullptr
                              : (OBF ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(a.add) + abase) : a.add + abase);
_Pragma("unroll")
    for (int i = 0; i < 16; ++i) {
      const bool vok = ((okrow >> (i >> 2)) & (okcol >> (i & 3)) & 1u) != 0;
      const unsigned ooff = oyo[i >> 2] + oxo[i & 3];
      const unsigned aoff = ayo[i >> 2] + axo[i & 3];
_Pragma("unroll")
      for (int nb = 0; nb < NB; ++nb) {
        const int col = nb * 32 + r;
        const bool live = vok && col < N;
        float val = acc[nb][i] + bias[nb];
        // fused add / accumulate operands share the output's storage type (host-checked)
        if (a.add) val += nl_apply(live ? ld1_t<OBF>(addp, aoff + col) : 0.f, asc[nb], ash[nb], a.tadd.relu);
        if (a.accumulate && live) val += ld1_t<OBF>(outp, ooff + col);
        if constexpr (OBF) {
          // bf16 rows: neighbouring channel lanes pair up, the even one stores both as one dword (a 2-byte store per
          // lane is a partial-dword write: ~12x the time per byte of a full store)
          const float other = __shfl_xor(val, 1, 64);
          if (live && !(r & 1)) {
            if (col + 1 < N) reinterpret_cast<unsigned int*>(reinterpret_cast<unsigned short*>(outp) + ooff + col)[0] = f32x2_to_bf16x2(val, other);
            else st1_t<true>(outp, ooff + col, val);
          }
        } else {
          if (live) outp[ooff + col] = val;
        }
        if (live) { ssum[nb] += val; ssq[nb] += val * val; }
      }
    }
  });
  if (a.stats != nullptr) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const float s0 = ssum[nb] + __shfl_xor(ssum[nb], 32, 64), s1 = ssq[nb] + __shfl_xor(ssq[nb], 32, 64);
      if (h == 0) { red[0][wave][nb * 32 + r] = s0; red[1][wave][nb * 32 + r] = s1; }
    }
    __syncthreads();
    if (tid < 32 * NB && tid < N) {
      const float s0 = (red[0][0][tid] + red[0][1][tid]) + (red[0][2][tid] + red[0][3][tid]);
      const float s1 = (red[1][0][tid] + red[1][1][tid]) + (red[1][2][tid] + red[1][3][tid]);
      const long long row = (long long)n * a.tiles_per_n + tile_in_n;
      a.stats[(row * 2 + 0) * N + tid] = s0;
      a.stats[(row * 2 + 1) * N + tid] = s1;
    }
  }
}

bool chan_applicable(const mmtta_conv_desc* d, const mmtta_tensor* x, const mmtta_tensor* y) {
  if (!(d->op == MMTTA_CONV_FWD || d->op == MMTTA_CONVT_DGRAD) || d->ksize != 3) return false;
  const int K = x->c, N = y->c;
  // N = 64 (one voxel per pass) measured slower than the fp32 implicit GEMM (109 vs 80 us on the 64->3 up-convolution's
  // input gradient: 27 LDS reads per voxel and wave); kept for N = 32, two voxels per pass
  // N = 64 only on the matrix cores (bf16 mode): the input gradient of the 64->R up-convolution
  // a one-channel slice of a wider tensor (modality m of the network input) need not start on a 16-byte group: the kernels
  // load the group it sits in and pick its float (CArgs::koff)
  const bool slice1 = K == 1 && x->sc == 1 && x->sw % 4 == 0 && x->sh % 4 == 0 && x->sd % 4 == 0 && x->sn % 4 == 0 &&
                      ((uintptr_t)x->ptr) % 4 == 0;
  // the kernels address with 32-bit element offsets inside a batch item (input) / a z-slice (output)
  const bool small = (long long)x->d * x->sd < (1LL << 31) && (long long)(y->h + 4) * y->sh < (1LL << 31);
  // bf16-stored gathered tensor (8-byte voxels): the matrix-core kernel
  if (is_bf16(x))
    return d->dtype == MMTTA_BF16 && K <= 4 && ((N == 32 && K >= 2) || N == 64) && x->sc == 1 && x->sw % 4 == 0 && x->sh % 4 == 0 &&
           x->sd % 4 == 0 && x->sn % 4 == 0 && ((uintptr_t)x->ptr) % 8 == 0 && small;
  return K <= 4 && (N == 32 || (N == 64 && d->dtype == MMTTA_BF16)) && (aligned16(x) || slice1) && x->sw >= 4 && small;
}

int chan_tiles_per_n(const mmtta_tensor* y) { return ((y->d + 3) / 4) * ((y->h + 3) / 4) * ((y->w + 7) / 8); }

template <int S, int KI, bool HAS_T>
static void launch_chan_n(const CArgs& a, int N, int blocks, hipStream_t s) {
  if (N == 64) hipLaunchKernelGGL((direct_chan_kernel<S, KI, 64, HAS_T>), dim3(blocks), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((direct_chan_kernel<S, KI, 32, HAS_T>), dim3(blocks), dim3(256), 0, s, a);
}

template <int S, bool HAS_T>
static void launch_chan_k(const CArgs& a, int K, int N, int blocks, hipStream_t s) {
  switch (K) {
    case 1: launch_chan_n<S, 1, HAS_T>(a, N, blocks, s); break;
    case 2: launch_chan_n<S, 2, HAS_T>(a, N, blocks, s); break;
    case 3: launch_chan_n<S, 3, HAS_T>(a, N, blocks, s); break;
    default: launch_chan_n<S, 4, HAS_T>(a, N, blocks, s); break;
  }
}

template <int S, int KI, bool HAS_T>
static void launch_chan_mfma_n(const CArgs& a, int N, int blocks, hipStream_t s) {
  if (a.in.bf) {                     // bf16-stored gathered tensor (network input; thin gradients)
    if (N > 32) hipLaunchKernelGGL((chan_mfma_kernel<S, KI, 2, HAS_T, true>), dim3(blocks), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((chan_mfma_kernel<S, KI, 1, HAS_T, true>), dim3(blocks), dim3(256), 0, s, a);
    return;
  }
  if (N > 32) hipLaunchKernelGGL((chan_mfma_kernel<S, KI, 2, HAS_T>), dim3(blocks), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((chan_mfma_kernel<S, KI, 1, HAS_T>), dim3(blocks), dim3(256), 0, s, a);
}

template <int S, bool HAS_T>
static void launch_chan_mfma(const CArgs& a, int K, int N, int blocks, hipStream_t s) {
  switch (K) {
    case 1: launch_chan_mfma_n<S, 1, HAS_T>(a, N, blocks, s); break;
    case 2: launch_chan_mfma_n<S, 2, HAS_T>(a, N, blocks, s); break;
    case 3: launch_chan_mfma_n<S, 3, HAS_T>(a, N, blocks, s); break;
    default: launch_chan_mfma_n<S, 4, HAS_T>(a, N, blocks, s); break;
  }
}

int chan_conv_run(const mmtta_conv_desc* d, const mmtta_tensor* x, const mmtta_norm_on_load* x_norm, const void* packed, int Kp,
                  int Np, const float* bias, const mmtta_conv_epilogue* epi, const mmtta_tensor* y, int accumulate, float* stats,
                  const PSets& sets, hipStream_t stream) {
  CArgs a;
  a.ps = sets;
  a.in = tv(x); a.tin = nl(x_norm); a.out = tv(y);
  a.koff = is_bf16(x) ? 0 : (int)((((uintptr_t)x->ptr) % 16) / 4);
  MMTTA_CHECK(a.koff == 0 || x->c == 1, MMTTA_ERR_UNSUPPORTED, "thin-K conv: only a one-channel slice may start inside a 16-byte group");
  a.in.p -= a.koff;
  a.w = (const float*)packed; a.Kp = Kp; a.Np = Np; a.bias = bias;
  a.add = nullptr; a.asn = a.asd = a.ash = a.asw = 0; a.tadd = nl(nullptr); a.add_bf = 0;
  if (epi && epi->add) {
    const mmtta_tensor* ad = epi->add;
    MMTTA_CHECK(ad->ptr && is_cl(ad) && ad->n == y->n && ad->c == y->c && ad->d == y->d && ad->h == y->h && ad->w == y->w,
                MMTTA_ERR_INVALID, "conv: epilogue `add` must be channels-last with the shape of y");
    a.add = (const float*)ad->ptr; a.asn = ad->sn; a.asd = ad->sd; a.ash = ad->sh; a.asw = ad->sw;
    a.tadd = nl(&epi->add_norm);
    a.add_bf = is_bf16(ad) ? 1 : 0;
    MMTTA_CHECK((long long)(ad->h + 4) * ad->sh < (1LL << 31), MMTTA_ERR_UNSUPPORTED, "thin-K conv: epilogue `add` slice beyond 2^31 elements");
  }
  // the <= 4-channel gathered tensor is fp32 (gradients, fp32 precision) or - the network input of bf16 precision - bf16
  // with 8-byte voxels, on the matrix-core path; the 32 / 64-channel result may be bf16-stored, on that path only
  // (a bf16-stored result - the deep-fusion stems under method.storage: bf16 - takes the matrix-core kernel for one input
  // channel as well: the VALU kernel writes fp32 only)
  const bool mfma_path = d->dtype == MMTTA_BF16 && (x->c >= 2 || y->c > 32 || is_bf16(y));
  MMTTA_CHECK(is_f32(x) || mfma_path, MMTTA_ERR_UNSUPPORTED, "thin-K conv: a bf16-stored gathered tensor needs the matrix-core path");
  MMTTA_CHECK(mfma_path || (is_f32(y) && !a.add_bf), MMTTA_ERR_UNSUPPORTED, "thin-K conv (VALU path): fp32-stored tensors only");
  MMTTA_CHECK(a.add == nullptr || (a.add_bf != 0) == is_bf16(y), MMTTA_ERR_UNSUPPORTED, "thin-K conv: the fused add must share the output's storage type");
  a.accumulate = accumulate; a.stats = stats;
  a.tz = (y->d + 3) / 4; a.ty = (y->h + 3) / 4; a.tx = (y->w + 7) / 8;
  a.tiles_per_n = a.tz * a.ty * a.tx;
  const int blocks = a.tiles_per_n * y->n;
  const bool has_t = a.tin.mean != nullptr || a.tin.scale != nullptr;
  const int S = d->op == MMTTA_CONV_FWD ? d->stride : 2;       // CONVT_DGRAD: stride-2 gather
  // one input channel (the single-modality stems of the deep-fusion net) is 27 FMAs per output: the VALU kernel wins there
  if (mfma_path) {
    if (S == 1) { if (has_t) launch_chan_mfma<1, true>(a, x->c, y->c, blocks, stream); else launch_chan_mfma<1, false>(a, x->c, y->c, blocks, stream); }
    else { if (has_t) launch_chan_mfma<2, true>(a, x->c, y->c, blocks, stream); else launch_chan_mfma<2, false>(a, x->c, y->c, blocks, stream); }
    return launch_status("thin-K conv (bf16 MFMA)");
  }
  if (S == 1) { if (has_t) launch_chan_k<1, true>(a, x->c, y->c, blocks, stream); else launch_chan_k<1, false>(a, x->c, y->c, blocks, stream); }
  else { if (has_t) launch_chan_k<2, true>(a, x->c, y->c, blocks, stream); else launch_chan_k<2, false>(a, x->c, y->c, blocks, stream); }
  return launch_status("direct conv (lanes along N)");
}

// ------------------------------------------------------------------ 1x1x1 head, K = 32 / 64 -> N <= 4, voxel-dense tensors
// The final 1x1x1 convolution of the deep-fusion net (32 -> 3 at full resolution): 3 x 32 FMAs per voxel, pure streaming -
// 64 bytes of bf16 (128 of fp32) in, 16 out.  The lanes-along-K kernel (8 lanes share a voxel, halving exchange at the end)
// took 1.2 ms per group of 8 volumes at 128^3 against 0.2 ms of HBM time.  Here a thread owns a voxel: K / 8 16-byte loads
// (all issued before the first use), the K x 4 weights read from LDS as broadcasts, one 16-byte store.
template <int K, bool XBF>
__global__ __launch_bounds__(256) void pointwise_head_kernel(DArgs a) {
  __shared__ float4 wl[K];
  const int n = blockIdx.y;
  a.w = pset_packed(a.ps, a.w, n); a.bias = pset_bias(a.ps, a.bias, n);
  for (int i = threadIdx.x; i < K; i += 256) wl[i] = *reinterpret_cast<const float4*>(a.w + 4 * i);      // [K][4]
  float b[4] = {0.f, 0.f, 0.f, 0.f};
  if (a.bias) {
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = j < a.N ? a.bias[j] : 0.f;
  }
  __syncthreads();
  const long long dhw = (long long)a.in.d * a.in.h * a.in.w;
  const float* inb = XBF ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(a.in.p) + (long long)n * a.in.sn)
                         : a.in.p + (long long)n * a.in.sn;
  float* outb = a.out.p + (long long)n * a.out.sn;
  for (long long v = blockIdx.x * 256LL + threadIdx.x; v < dhw; v += (long long)gridDim.x * 256) {
    Oct8<XBF> raw[K / 8];
#pragma unroll
    for (int c = 0; c < K / 8; ++c) raw[c] = oct8_ld<XBF>(inb, (unsigned)(v * a.in.sw) + 8 * c, (unsigned)(v * a.in.sw) + 8 * c + 4);
    float o[4] = {b[0], b[1], b[2], b[3]};
#pragma unroll
    for (int c = 0; c < K / 8; ++c) {
      float x8[8];
      oct8_f8(raw[c], x8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float4 w4 = wl[8 * c + e];
        o[0] = fmaf(x8[e], w4.x, o[0]); o[1] = fmaf(x8[e], w4.y, o[1]); o[2] = fmaf(x8[e], w4.z, o[2]); o[3] = fmaf(x8[e], w4.w, o[3]);
      }
    }
    *reinterpret_cast<float4*>(outb + v * a.out.sw) = make_float4(o[0], a.N > 1 ? o[1] : 0.f, a.N > 2 ? o[2] : 0.f, a.N > 3 ? o[3] : 0.f);
  }
}

// ------------------------------------------------------------------ 1x1x1, K <= 4 -> N (multiple of 4), no statistics
// The input gradient of a 1x1 head (deep-fusion final_conv R->32 at full resolution): 3 FMAs per output element, pure
// streaming.  A thread owns 4 output channels of one voxel: one 16-byte load of the voxel's K inputs, one 16-byte
// store.  (On the fp32 implicit GEMM this layer took 350 us at 128^3.)
struct PArgs {
  TV in, out;
  const float* w; int Kp, Np;      // implicit-GEMM fp32 image [1][Kp][Np]
  const float* bias;
  int accumulate;
  PSets ps;                        // per-item parameter sets (the batch item varies per thread here)
};

template <int KI, bool OBF>
__global__ __launch_bounds__(256) void pointwise_small_k_kernel(PArgs a) {
  const int NV = a.out.c / 4;
  const long long dhw = (long long)a.out.d * a.out.h * a.out.w;
  const long long total = (long long)a.out.n * dhw * NV;
  const bool dense = a.in.sh == (long long)a.in.w * a.in.sw && a.in.sd == (long long)a.in.h * a.in.sh &&
                     a.out.sh == (long long)a.out.w * a.out.sw && a.out.sd == (long long)a.out.h * a.out.sh;
  const bool sets = psets_on(a.ps);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int n0 = (int)(i % NV) * 4;
    long long ioff, ooff;
    int nb;
    if (dense) {                                   // dense voxel order on both sides: no coordinate arithmetic
      const long long v = i / NV, n = v / dhw, r = v - n * dhw;
      ioff = n * a.in.sn + r * a.in.sw;
      ooff = n * a.out.sn + r * a.out.sw;
      nb = (int)n;
    } else {
      int n, z, y, x;
      vox_decompose(a.out, i / NV, n, z, y, x);
      ioff = vox_addr(a.in, n, z, y, x);
      ooff = vox_addr(a.out, n, z, y, x);
      nb = n;
    }
    const float* wn = a.w;
    const float* bn = a.bias;
    if (sets) { wn = pset_packed(a.ps, a.w, nb); bn = pset_bias(a.ps, a.bias, nb); }
    const float4 x4 = *reinterpret_cast<const float4*>(a.in.p + ioff);
    const float xs[4] = {x4.x, x4.y, x4.z, x4.w};
    float o[4] = {0.f, 0.f, 0.f, 0.f};
    if (bn) {
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = bn[n0 + j];
    }
#pragma unroll
    for (int k = 0; k < KI; ++k) {
      const float4 w4 = *reinterpret_cast<const float4*>(wn + (long long)k * a.Np + n0);
      o[0] = fmaf(xs[k], w4.x, o[0]); o[1] = fmaf(xs[k], w4.y, o[1]); o[2] = fmaf(xs[k], w4.z, o[2]); o[3] = fmaf(xs[k], w4.w, o[3]);
    }
    if (a.accumulate) {
      const float4 t = ld4_any(a.out.p, ooff + n0, OBF);
      o[0] += t.x; o[1] += t.y; o[2] += t.z; o[3] += t.w;
    }
    st4_any(a.out.p, ooff + n0, make_float4(o[0], o[1], o[2], o[3]), OBF);
  }
}

bool pointwise_small_applicable(const mmtta_conv_desc* d, const mmtta_tensor* x, const mmtta_tensor* y, const float* stats,
                                const mmtta_conv_epilogue* epi, const mmtta_norm_on_load* x_norm) {
  if (d->ksize != 1 || d->stride != 1 || stats != nullptr || (epi && epi->add)) return false;
  if (x_norm && (x_norm->mean || x_norm->scale)) return false;
  if (!(d->op == MMTTA_CONV_FWD || d->op == MMTTA_CONV_DGRAD)) return false;
  return x->c <= 4 && is_f32(x) && y->c % 4 == 0 && y->c >= 4 && aligned16(x) && x->sw >= 4 &&
         (is_bf16(y) ? (y->sc == 1 && y->sw % 4 == 0 && y->sh % 4 == 0 && y->sd % 4 == 0 && y->sn % 4 == 0 && ((uintptr_t)y->ptr) % 8 == 0)
                     : aligned16(y));
}

int pointwise_small_run(const mmtta_tensor* x, const void* packed, int Kp, int Np, const float* bias, const mmtta_tensor* y,
                        int accumulate, const PSets& sets, hipStream_t stream) {
  PArgs a;
  a.in = tv(x); a.out = tv(y); a.w = (const float*)packed; a.Kp = Kp; a.Np = Np; a.bias = bias; a.accumulate = accumulate;
  a.ps = sets;
  const long long total = (long long)y->n * y->d * y->h * y->w * (y->c / 4);
  long long blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
#define MMTTA_PW(KI) \
  do { if (is_bf16(y)) hipLaunchKernelGGL((pointwise_small_k_kernel<KI, true>), dim3((unsigned)blocks), dim3(256), 0, stream, a); \
       else hipLaunchKernelGGL((pointwise_small_k_kernel<KI, false>), dim3((unsigned)blocks), dim3(256), 0, stream, a); } while (0)
  switch (x->c) {           // (the output may be a bf16-stored gradient: method.grad_storage)
    case 1: MMTTA_PW(1); break;
    case 2: MMTTA_PW(2); break;
    case 3: MMTTA_PW(3); break;
    default: MMTTA_PW(4); break;
  }
#undef MMTTA_PW
  return launch_status("1x1 conv (small K)");
}

// 0: thread per voxel (any shape); 1: lanes along K (K = 32 or 64); 2: row kernel (K <= 4, k3 s1);
// 3: LDS-staged stride-2 up-convolution (ConvTranspose3d forward, K = 32 or 64); 4: variant 2's shapes on the 4x4x4
// matrix tiles (bf16 precision, MMTTA_OPT_THIN_MFMA)
static int direct_variant(const mmtta_conv_desc* d, const mmtta_tensor* x) {
  int K, N;
  direct_dims(d, K, N);
  if (!aligned16(x) || (long long)x->w * x->sw * 4 >= (1LL << 31)) return 0;
  if ((K == 32 || K == 64) && d->op == MMTTA_CONVT_FWD && d->ksize == 3 && d->stride == 2 &&
      (long long)x->d * x->sd < (1LL << 31)) {               // 32-bit offsets inside a batch item
    // bf16 precision: the 2x2x2 gather GEMM (upconv8_kernel) when the input admits its 8-channel items
    const int q = is_bf16(x) ? 8 : 4;
    if (d->dtype == MMTTA_BF16 && x->sw % q == 0 && x->sh % q == 0 && x->sd % q == 0 && x->sn % q == 0) return 5;
    return 3;
  }
  if (K == 32 || K == 64) return 1;
  if (K <= 4 && d->stride == 1 && d->ksize == 3) return (d->dtype == MMTTA_BF16 && g_thin_mfma && N <= 4) ? 4 : 2;
  return 0;
}

static long long direct_units(const mmtta_conv_desc* d, int variant, const mmtta_tensor* y) {
  if (variant == 2) return (long long)y->d * ((y->h + 1) / 2) * ((y->w + 63) / 64);     // pairs of rows
  const bool s2t = (d->op == MMTTA_CONVT_FWD || d->op == MMTTA_CONV_DGRAD) && d->stride == 2;
  const int xstep = s2t ? 2 : 1;
  const int rowlen = (y->w + xstep - 1) / xstep;
  return (long long)y->d * y->h * xstep * ((rowlen + 63) / 64);
}

// workgroups per batch item = statistics rows per batch item
int direct_blocks_per_n(const mmtta_conv_desc* d, const mmtta_tensor* x, const mmtta_tensor* y) {
  const int v = direct_variant(d, x);
  if (v == 0) {
    const long long dhw = (long long)y->d * y->h * y->w;
    return (int)((dhw + 255) / 256);
  }
  if (v == 3) return x->d * x->h * ((x->w + 63) / 64);       // one workgroup per (input row pair, 64-voxel chunk)
  if (v == 5) return ((x->d + 3) / 4) * ((x->h + 3) / 4) * ((x->w + 7) / 8);      // one statistics row per coarse 4 x 4 x 8 tile
  if (v == 4) {                                    // one workgroup per TZ x 8 x 64 tile (MMTTA_OPT_THIN_MFMA = 2: TZ = 4)
    const int tz = g_thin_mfma == 3 ? 2 : g_thin_mfma == 2 ? 4 : 8;
    return ((y->d + tz - 1) / tz) * ((y->h + 7) / 8) * ((y->w + 63) / 64);
  }
  // grid-stride kernels: enough workgroups to fill the chip a few times over, never more than the work
  const long long want = (direct_units(d, v, y) + 3) / 4;
  const long long cap = v == 1 ? 1024 : 2048;
  return (int)(want < cap ? want : cap);
}

template <int KL, bool HAS_T>
static void launch_klane(const DArgs& a, int n, size_t lds, hipStream_t stream) {
  const dim3 grid(a.blocks_per_n, n), block(256);
  if (a.in.bf) {
    switch (a.N) {
      case 1: hipLaunchKernelGGL((direct_klane_kernel<KL, 1, HAS_T, true>), grid, block, lds, stream, a); break;
      case 2: hipLaunchKernelGGL((direct_klane_kernel<KL, 2, HAS_T, true>), grid, block, lds, stream, a); break;
      case 3: hipLaunchKernelGGL((direct_klane_kernel<KL, 3, HAS_T, true>), grid, block, lds, stream, a); break;
      default: hipLaunchKernelGGL((direct_klane_kernel<KL, 4, HAS_T, true>), grid, block, lds, stream, a); break;
    }
    return;
  }
  switch (a.N) {
    case 1: hipLaunchKernelGGL((direct_klane_kernel<KL, 1, HAS_T>), grid, block, lds, stream, a); break;
    case 2: hipLaunchKernelGGL((direct_klane_kernel<KL, 2, HAS_T>), grid, block, lds, stream, a); break;
    case 3: hipLaunchKernelGGL((direct_klane_kernel<KL, 3, HAS_T>), grid, block, lds, stream, a); break;
    default: hipLaunchKernelGGL((direct_klane_kernel<KL, 4, HAS_T>), grid, block, lds, stream, a); break;
  }
}

template <int KI, bool HAS_T>
static void launch_row_n(const DArgs& a, int n, hipStream_t stream) {
  const dim3 grid(a.blocks_per_n, n), block(256);
  switch (a.N) {
    case 1: hipLaunchKernelGGL((direct_row_kernel<KI, 1, HAS_T>), grid, block, 0, stream, a); break;
    case 2: hipLaunchKernelGGL((direct_row_kernel<KI, 2, HAS_T>), grid, block, 0, stream, a); break;
    case 3: hipLaunchKernelGGL((direct_row_kernel<KI, 3, HAS_T>), grid, block, 0, stream, a); break;
    default: hipLaunchKernelGGL((direct_row_kernel<KI, 4, HAS_T>), grid, block, 0, stream, a); break;
  }
}

// ------------------------------------------------------------------ the same up-convolution on the matrix cores
// `bf16` precision mode.  Per tap the work is [R x K] x [K x 64 voxels]: far too thin for a tiled GEMM, but the fp32
// VALU form above is bound by the latency of its LDS + scalar operand fetches (both count on lgkmcnt, so every weight
// use waits for all LDS reads in flight).  Here the weights are the A operand (rows = output channel, 32 rows of which
// R <= 4 are non-zero) and the staged activations the B operand (columns = 32 voxels of the row) of
// v_mfma_f32_32x32x16_bf16: lanes 0-31 end with accumulator rows 0..3 = the R outputs of "their" voxel, exactly what the
// shared epilogue wants, and one ds_read_b128 per lane feeds 32 x 32 x 16 MACs.  Staging, parity classes and epilogue
// are those of direct_upconv_kernel; the weights are rounded to bf16 into a compact LDS image [tap][K/16][half][4 rows].

template <int K, int NO, bool HAS_T>
__global__ __launch_bounds__(256) void upconv_mfma_kernel(DArgs a) {
  extern __shared__ float lds[];
  constexpr int VS = K / 2 + 4;                // LDS voxel stride (dwords): conflict-free ds_read_b128 across voxels
  constexpr int XV = 65;
  constexpr int KS = K / 16;                   // MFMA k-steps per tap
  uint4* wimg = reinterpret_cast<uint4*>(lds + 4 * XV * VS);          // [27][KS][2][4] x 8 bf16
  float* red = lds + 4 * XV * VS + 27 * KS * 2 * 4 * 4;
  const int n = blockIdx.y;
  a.w = pset_packed(a.ps, a.w, n); a.bias = pset_bias(a.ps, a.bias, n);      // this batch item's parameter set
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int chunks = (a.in.w + 63) / 64;
  int b = blockIdx.x;
  const int chunk = b % chunks; b /= chunks;
  const int iy0 = b % a.in.h;
  const int iz0 = b / a.in.h;
  const int ix0 = chunk * 64;
  {  // ---- stage the four input rows (normalised, rounded to bf16)
    constexpr int CG = K / 4;
    constexpr int NIT = (4 * XV * CG + 255) / 256;
    const int cg = tid % CG;
    float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
    if (HAS_T) nl_coeff_vec<4>(a.tin, n, K, cg * 4, sc, sh);
    // the four rows' offsets and validity are workgroup-uniform (scalar); a staged voxel adds its clamped x offset: 32-bit
    // elements from the batch item (host-checked), no 64-bit vector arithmetic
    const long long inbo = (long long)n * a.in.sn + cg * 4;      // element offset (the input may be bf16-stored)
    unsigned rowoff[4], rowok = 0;
#pragma unroll
    for (int row = 0; row < 4; ++row) {
      const int iz = iz0 + (row >> 1), iy = iy0 + (row & 1);
      rowoff[row] = (unsigned)min(iz, a.in.d - 1) * (unsigned)a.in.sd + (unsigned)min(iy, a.in.h - 1) * (unsigned)a.in.sh;
      rowok |= (iz < a.in.d && iy < a.in.h ? 1u : 0u) << row;
    }
    const unsigned sw = (unsigned)a.in.sw;
    float4 raw[NIT];
    unsigned okm = 0;
    MMTTA_BF_DISPATCH(a.in.bf, INBF, {
      const float* inb = INBF ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(a.in.p) + inbo) : a.in.p + inbo;
_Pragma("unroll")
      for (int q = 0; q < NIT; ++q) {
        const int vs = min(tid / CG + q * (256 / CG), 4 * XV - 1);
        const int row = vs / XV, xl = vs % XV;
        const unsigned ro = row == 0 ? rowoff[0] : row == 1 ? rowoff[1] : row == 2 ? rowoff[2] : rowoff[3];
        okm |= (((rowok >> row) & 1u) != 0 && ix0 + xl < a.in.w ? 1u : 0u) << q;
        raw[q] = ld4_t<INBF>(inb, (long long)(ro + (unsigned)min(ix0 + xl, a.in.w - 1) * sw));
      }
    });
#pragma unroll
    for (int q = 0; q < NIT; ++q) {
      const int vs = tid / CG + q * (256 / CG);
      if (vs < 4 * XV) {
        const bool ok = (okm >> q) & 1u;
        float4 v;
        v.x = ok ? (HAS_T ? nl_apply(raw[q].x, sc[0], sh[0], a.tin.relu) : raw[q].x) : 0.f;
        v.y = ok ? (HAS_T ? nl_apply(raw[q].y, sc[1], sh[1], a.tin.relu) : raw[q].y) : 0.f;
        v.z = ok ? (HAS_T ? nl_apply(raw[q].z, sc[2], sh[2], a.tin.relu) : raw[q].z) : 0.f;
        v.w = ok ? (HAS_T ? nl_apply(raw[q].w, sc[3], sh[3], a.tin.relu) : raw[q].w) : 0.f;
        uint2 pk;
        pk.x = __builtin_bit_cast(unsigned int, __builtin_convertvector((float2_t){v.x, v.y}, bf16x2_t));
        pk.y = __builtin_bit_cast(unsigned int, __builtin_convertvector((float2_t){v.z, v.w}, bf16x2_t));
        *reinterpret_cast<uint2*>(lds + vs * VS + cg * 2) = pk;
      }
    }
  // ---- weight image: entry ((tap * KS + s) * 2 + half) * 4 + row holds w[tap][s*16 + half*8 + 0..7][row] as bf16.
  // One thread per (tap, s, half): eight contiguous 16-byte loads (8 channels x 4 rows), four entries out.
  if (tid < 27 * KS * 2) {
    const float4* wp = reinterpret_cast<const float4*>(a.w) + tid * 8;      // (tap*K + s*16 + half*8) = tid * 8
    float4 wv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) wv[j] = wp[j];
    auto comp = [](const float4& v, int r) { return r == 0 ? v.x : (r == 1 ? v.y : (r == 2 ? v.z : v.w)); };
#pragma unroll
    for (int row = 0; row < 4; ++row) {
      uint4 pk = make_uint4(0u, 0u, 0u, 0u);
      if (row < NO) {
        pk.x = __builtin_bit_cast(unsigned int, __builtin_convertvector((float2_t){comp(wv[0], row), comp(wv[1], row)}, bf16x2_t));
        pk.y = __builtin_bit_cast(unsigned int, __builtin_convertvector((float2_t){comp(wv[2], row), comp(wv[3], row)}, bf16x2_t));
        pk.z = __builtin_bit_cast(unsigned int, __builtin_convertvector((float2_t){comp(wv[4], row), comp(wv[5], row)}, bf16x2_t));
        pk.w = __builtin_bit_cast(unsigned int, __builtin_convertvector((float2_t){comp(wv[6], row), comp(wv[7], row)}, bf16x2_t));
      }
      wimg[tid * 4 + row] = pk;
    }
  }
  }
  __syncthreads();
  const int h = lane >> 5, r = lane & 31;
  const bool wrow = r < 4;                     // lanes that hold a (possibly zero) weight row
  float ssum[NO], ssq[NO];
#pragma unroll
  for (int c = 0; c < NO; ++c) { ssum[c] = 0.f; ssq[c] = 0.f; }
  // (Dealing the 16 (class, half-row) jobs evenly over the waves was measured slower, 84 vs 70 us at 64^3 -> 128^3:
  // the kernel is bound by staging and the scattered epilogue, not by the MFMA pipe.)
  const int pz = wave >> 1, py = wave & 1;     // a wave owns one (z, y) parity; both x parities -> adjacent stores
  float vals[2][NO];
#pragma unroll
  for (int px = 0; px < 2; ++px) {
    ufloat16 acc[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }
    for (int tz = 0; tz <= pz; ++tz) {
      const int kz = pz == 0 ? 1 : (tz == 0 ? 2 : 0);
      for (int ty = 0; ty <= py; ++ty) {
        const int ky = py == 0 ? 1 : (ty == 0 ? 2 : 0);
        for (int tx = 0; tx <= px; ++tx) {
          const int kx = px == 0 ? 1 : (tx == 0 ? 2 : 0);
          const int tap = (kz * 3 + ky) * 3 + kx;
          const float* xrow = lds + ((tz * 2 + ty) * XV + tx + r) * VS + h * 4;
          const uint4* wt = wimg + (tap * KS * 2 + h) * 4 + (wrow ? r : 0);
          uint4 wq[KS], x0[KS], x1[KS];
#pragma unroll
          for (int s2 = 0; s2 < KS; ++s2) {
            wq[s2] = wt[s2 * 8];
            x0[s2] = *reinterpret_cast<const uint4*>(xrow + s2 * 8);
            x1[s2] = *reinterpret_cast<const uint4*>(xrow + 32 * VS + s2 * 8);
          }
#pragma unroll
          for (int s2 = 0; s2 < KS; ++s2) {
            if (!wrow) wq[s2] = make_uint4(0u, 0u, 0u, 0u);
            const ubf16x8 af = __builtin_bit_cast(ubf16x8, wq[s2]);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(ubf16x8, x0[s2]), acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(ubf16x8, x1[s2]), acc[1], 0, 0, 0);
          }
        }
      }
    }
    // lanes 0..31 hold accumulator rows 0..3 = the outputs of voxels r (acc[0]) and 32 + r (acc[1]); the second set
    // moves to lanes 32..63 so that one full-width epilogue serves the 64 voxels of the row
#pragma unroll
    for (int c = 0; c < NO; ++c) {
      const float upper = __shfl(acc[1][c], r);
      vals[px][c] = h ? upper : acc[0][c];
    }
  }
  {
    const int oz = 2 * iz0 + pz, oy = 2 * iy0 + py;
    const int ixl = ix0 + lane;
    if (ixl < a.in.w && oz < a.out.d && oy < a.out.h) {
      if (2 * ixl < a.out.w) direct_epilogue<NO>(a, n, oz, oy, 2 * ixl, vals[0], ssum, ssq);
      if (2 * ixl + 1 < a.out.w) direct_epilogue<NO>(a, n, oz, oy, 2 * ixl + 1, vals[1], ssum, ssq);
    }
  }
  direct_stats<NO>(a, n, ssum, ssq, red);
}

template <int K, bool HAS_T>
static void launch_upconv_mfma(const DArgs& a, int n, hipStream_t stream) {
  const dim3 grid(a.blocks_per_n, n), block(256);
  const size_t lds = ((size_t)4 * 65 * (K / 2 + 4) + 27 * (K / 16) * 2 * 4 * 4 + 32) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)upconv_mfma_kernel<K, 1, HAS_T>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    (void)hipFuncSetAttribute((const void*)upconv_mfma_kernel<K, 2, HAS_T>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    (void)hipFuncSetAttribute((const void*)upconv_mfma_kernel<K, 3, HAS_T>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    (void)hipFuncSetAttribute((const void*)upconv_mfma_kernel<K, 4, HAS_T>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    attr_set = true;
  }
  switch (a.N) {
    case 1: hipLaunchKernelGGL((upconv_mfma_kernel<K, 1, HAS_T>), grid, block, lds, stream, a); break;
    case 2: hipLaunchKernelGGL((upconv_mfma_kernel<K, 2, HAS_T>), grid, block, lds, stream, a); break;
    case 3: hipLaunchKernelGGL((upconv_mfma_kernel<K, 3, HAS_T>), grid, block, lds, stream, a); break;
    default: hipLaunchKernelGGL((upconv_mfma_kernel<K, 4, HAS_T>), grid, block, lds, stream, a); break;
  }
}

// ------------------------------------------------------------------ the up-convolution as ONE 2x2x2 gather GEMM (round 3)
// ConvTranspose3d K -> R (k3 s2 p1 op1, R <= 4): the 8 fine voxels 2g + p of a coarse voxel g need exactly the 8 coarse
// neighbours g + d, d in {0,1}^3, and every one of the 27 taps belongs to ONE (offset d, parity p) pair.  So
//     out'[g][(p, co)] = sum_{d, ci} in[g + d][ci] * W'[d][ci][(p, co)]
// is a dense 8-offset convolution on the COARSE grid with N = 8 parities x 4 (padded) channels = 32 columns - one
// v_mfma_f32_32x32x16_bf16 column block - followed by a pixel shuffle.  The kernel above it (upconv_mfma_kernel, round 1)
// ran a workgroup per input row pair: every workgroup re-staged four rows, rebuilt the weight image from the fp32 taps and
// used 3 of 32 MFMA rows - 431 us for a group of 8 volumes at 64^3 -> 128^3 against 75 us of HBM time.  Here:
//   * W' is part of the PACKED image (written once per optimizer step by the pack kernels, bf16, already in B-fragment
//     order: entry ((d * K/16 + ks) * 64 + lane) = the lane's 8 k values of column lane & 31), copied into LDS once per
//     workgroup - workgroups are persistent over a contiguous range of tiles;
//   * a tile = 4 x 4 x 8 coarse voxels (one z-slice = one 32-row block per wave), its 5 x 5 x 9 halo box staged once as
//     bf16 [voxel][K] (norm + ReLU applied on the way, out-of-range voxels exact zeros);
//   * 8 offsets x K/16 MFMAs per wave and tile; the accumulator (row = coarse voxel, column = (parity, channel)) goes
//     through a wave-private LDS slab [2 fine z][8 fine y][16 fine x][4] laid out so that the 64 lanes hit 64 distinct banks,
//     and comes back as one float4 = one fine voxel per lane: 256-byte runs per output row, bias / fused add / accumulate
//     / statistics in the shared epilogue.
__device__ __forceinline__ void upconv8_slot(int tap, int& d, int& p) {
  // per axis: k = 1 -> parity 0 reads g; k = 0 -> parity 1 reads g + 1; k = 2 -> parity 1 reads g   (out = 2 g - 1 + k)
  const int kz = tap / 9, ky = (tap / 3) % 3, kx = tap % 3;
  p = ((kz != 1 ? 1 : 0) * 2 + (ky != 1 ? 1 : 0)) * 2 + (kx != 1 ? 1 : 0);
  d = ((kz == 0 ? 1 : 0) * 2 + (ky == 0 ? 1 : 0)) * 2 + (kx == 0 ? 1 : 0);
}
// halfword index of W'[d][k][(p, co)] in the fragment-ordered image
__device__ __forceinline__ int upconv8_index(int K, int d, int k, int col) {
  return (((d * (K >> 4) + (k >> 4)) * 64 + ((k >> 3) & 1) * 32 + col) << 3) + (k & 7);
}

template <int K, int NO, bool HAS_T, bool INBF>
__global__ __launch_bounds__(256, 2) void upconv8_kernel(DArgs a, int tiles_per_n, int tz, int ty, int tx) {
  constexpr int TZ = 4, TY = 4, TX = 8, BZ = TZ + 1, BY = TY + 1, BX = TX + 1;
  constexpr int VS = K + 8;                           // halfwords per staged voxel: 16 lanes of a ds_read_b128 -> 16 distinct slots
  constexpr int KS = K / 16, KC8 = K / 8;
  constexpr int BOXV = BZ * BY * BX;
  constexpr int NIT = (BOXV * KC8 + 255) / 256;       // 8-channel items per thread
  constexpr int SLAB_Z = 8 * 72 + 16, SLAB = 2 * SLAB_Z;   // floats of one wave's output slab (padded: conflict-free writes)
  constexpr int BOX_HW = (BOXV * VS + 7) / 8 * 8;
  extern __shared__ float lds[];
  unsigned short* lh = reinterpret_cast<unsigned short*>(lds);                 // box image, later the four output slabs
  uint4* wl = reinterpret_cast<uint4*>(lh + (BOX_HW > SLAB * 4 * 2 ? BOX_HW : SLAB * 4 * 2));   // W': 8 * KS * 64 fragments
  float* red = reinterpret_cast<float*>(wl + 8 * KS * 64);                     // 32 floats
  const int n = blockIdx.y;
  a.w = pset_packed(a.ps, a.w, n); a.bias = pset_bias(a.ps, a.bias, n);        // this batch item's parameter set
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, r = lane & 31;
  {  // W' of this batch item's set: behind the fp32 tap image [27][K][4]
    const uint4* src = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(a.w) + (size_t)27 * K * 16);
    for (int i = tid; i < 8 * KS * 64; i += 256) wl[i] = src[i];
  }
  // a thread's items i = tid + 256 j all carry the same channel octet (256 % KC8 == 0): one coefficient set per thread
  static_assert(256 % KC8 == 0, "items of a thread must share their channel octet");
  float sc[8], sh[8];
  if (HAS_T) nl_coeff_vec<8>(a.tin, n, K, (tid % KC8) * 8, sc, sh);
  const float relu_lo = (HAS_T && a.tin.relu) ? 0.f : -__builtin_inff();
  const float* inb = item_base<INBF>(a.in.p, n, a.in.sn);
  const unsigned isd = (unsigned)a.in.sd, ish = (unsigned)a.in.sh, isw = (unsigned)a.in.sw;
  long long tfirst, tlast;
  unit_range(tiles_per_n, xcd_contiguous_id(blockIdx.x, gridDim.x), gridDim.x, tfirst, tlast);
  // staging roles: a thread's box items are the same for every tile (decoded once); the loads of tile t + 1 are issued
  // right after tile t's image is complete and land under its MFMAs and epilogue (each tile used to start with a full
  // memory latency, two workgroups per CU to hide it)
  int pos[NIT];                                       // box voxel of item j: bz << 16 | by << 8 | bx
  const int cv = tid % KC8;
#pragma unroll
  for (int j = 0; j < NIT; ++j) {
    const int bv = min(tid + 256 * j, BOXV * KC8 - 1) / KC8;
    const int bz = bv / (BY * BX), brem = bv - bz * (BY * BX), by = brem / BX, bx = brem - by * BX;
    pos[j] = (bz << 16) | (by << 8) | bx;
  }
  Oct8<INBF> raw[NIT];
  unsigned okm = 0u;
  auto issue = [&](long long tile) {
    int t = (int)tile;
    const int txi = t % tx; t /= tx;
    const int tyi = t % ty;
    const int tzi = t / ty;
    const int gz0 = tzi * TZ, gy0 = tyi * TY, gx0 = txi * TX;
    okm = 0u;
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
      const int iz = gz0 + (pos[j] >> 16), iy = gy0 + ((pos[j] >> 8) & 255), ix = gx0 + (pos[j] & 255);
      okm |= ((iz < a.in.d && iy < a.in.h && ix < a.in.w) ? 1u : 0u) << j;
      const unsigned off = (unsigned)min(iz, a.in.d - 1) * isd + (unsigned)min(iy, a.in.h - 1) * ish + (unsigned)min(ix, a.in.w - 1) * isw + cv * 8;
      raw[j] = oct8_ld<INBF>(inb, off, off + 4);
    }
  };
  if (tfirst < tlast) issue(tfirst);
  for (long long tile = tfirst; tile < tlast; ++tile) {
    int t = (int)tile;
    const int txi = t % tx; t /= tx;
    const int tyi = t % ty;
    const int tzi = t / ty;
    const int gz0 = tzi * TZ, gy0 = tyi * TY, gx0 = txi * TX;
    // ---- the halo box of this tile (requested one tile ago): transform / mask / pack
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
      float v[8];
      oct8_f8(raw[j], v);
      if (HAS_T) {
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = fmaxf(fmaf(v[q], sc[q], sh[q]), relu_lo);
      }
      const unsigned m = ((okm >> j) & 1u) ? 0xffffffffu : 0u;
      uint4 pk;
      pk.x = f32x2_to_bf16x2(v[0], v[1]) & m; pk.y = f32x2_to_bf16x2(v[2], v[3]) & m;
      pk.z = f32x2_to_bf16x2(v[4], v[5]) & m; pk.w = f32x2_to_bf16x2(v[6], v[7]) & m;
      if (tid + 256 * j < BOXV * KC8)
        *reinterpret_cast<uint4*>(lh + ((((pos[j] >> 16) * BY + ((pos[j] >> 8) & 255)) * BX + (pos[j] & 255)) * VS + cv * 8)) = pk;
    }
    __syncthreads();
    if (tile + 1 < tlast) issue(tile + 1);
    // ---- 8 offsets x KS k-steps: one 32 x 32 block per wave (rows = the 4 x 8 coarse voxels of z-slice `wave`)
    ufloat16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    {
      const unsigned short* arow = lh + ((wave * BY + (r >> 3)) * BX + (r & 7)) * VS + 8 * h;
      const uint4* brow = wl + lane;
#pragma unroll
      for (int d = 0; d < 8; ++d) {
        const unsigned short* ad = arow + ((((d >> 2) & 1) * BY + ((d >> 1) & 1)) * BX + (d & 1)) * VS;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const uint4 af = *reinterpret_cast<const uint4*>(ad + ks * 16);
          const uint4 bf = brow[(d * KS + ks) * 64];
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ubf16x8, af), __builtin_bit_cast(ubf16x8, bf), acc, 0, 0, 0);
        }
      }
    }
    __syncthreads();                                   // the box image is dead: its space becomes the output slabs
    // ---- pixel shuffle through LDS.  Accumulator i of lane (h, r): coarse (y, x) = (i >> 2, (i & 3) + 4 h), column r =
    // (pz, py, px, co).  Slab address = pz * SLAB_Z + (2 y + py) * 72 + (2 x + px) * 4 + co: for one i the 64 lanes fall on
    // banks co + 4 px + 8 py + 16 pz + 32 h - all distinct.
    float* slab = lds + wave * SLAB;
    {
      const int pz = r >> 4, py = (r >> 3) & 1, px = (r >> 2) & 1, co = r & 3;
      float* sp = slab + pz * SLAB_Z + py * 72 + (px + 8 * h) * 4 + co;
#pragma unroll
      for (int i = 0; i < 16; ++i) sp[(i >> 2) * 144 + (i & 3) * 8] = acc[i];
    }
    __syncthreads();
    float ssum[NO], ssq[NO];
#pragma unroll
    for (int c = 0; c < NO; ++c) { ssum[c] = 0.f; ssq[c] = 0.f; }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int v = lane + 64 * q;                     // fine voxel of the slab: (fz, fy, fx) = (v >> 7, (v >> 4) & 7, v & 15)
      const int fz = v >> 7, fy = (v >> 4) & 7, fx = v & 15;
      const float4 o4 = *reinterpret_cast<const float4*>(slab + fz * SLAB_Z + fy * 72 + fx * 4);
      const int oz = 2 * (gz0 + wave) + fz, oy = 2 * gy0 + fy, ox = 2 * gx0 + fx;
      if (oz < a.out.d && oy < a.out.h && ox < a.out.w) {
        const float vals[4] = {o4.x, o4.y, o4.z, o4.w};
        direct_epilogue<NO>(a, n, oz, oy, ox, vals, ssum, ssq);
      }
    }
    // ---- statistics row of this tile (the following norm): lanes -> waves -> one row
    if (a.stats != nullptr) {
#pragma unroll
      for (int c = 0; c < NO; ++c) {
        const float s = wave_sum(ssum[c]), q2 = wave_sum(ssq[c]);
        if (lane == 0) { red[(0 * 4 + wave) * 4 + c] = s; red[(1 * 4 + wave) * 4 + c] = q2; }
      }
    }
    __syncthreads();                                   // slabs read, partial sums visible: the next tile may stage
    if (a.stats != nullptr && tid < NO) {
      const float s = red[0 * 4 + tid] + red[1 * 4 + tid] + red[2 * 4 + tid] + red[3 * 4 + tid];
      const float q2 = red[16 + 0 * 4 + tid] + red[16 + 1 * 4 + tid] + red[16 + 2 * 4 + tid] + red[16 + 3 * 4 + tid];
      const long long rrow = (long long)n * tiles_per_n + tile;
      a.stats[(rrow * 2 + 0) * a.N + tid] = s;
      a.stats[(rrow * 2 + 1) * a.N + tid] = q2;
    }
  }
}

template <int K, bool HAS_T, bool INBF>
static void launch_upconv8(const DArgs& a, int n, int tz, int ty, int tx, hipStream_t stream) {
  constexpr int VS = K + 8, BOX_HW = (225 * VS + 7) / 8 * 8, SLAB = 2 * (8 * 72 + 16);
  constexpr size_t lds = (size_t)(BOX_HW > SLAB * 8 ? BOX_HW : SLAB * 8) * 2 + (size_t)8 * (K / 16) * 64 * 16 + 32 * sizeof(float);
  const int tiles_per_n = tz * ty * tx;
  // persistent workgroups: each copies W' once and walks a contiguous range of tiles (an XCD sweeps one part of the volume)
  const dim3 grid(tiles_per_n < 128 ? tiles_per_n : 128, n), block(256);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)upconv8_kernel<K, 1, HAS_T, INBF>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipFuncSetAttribute((const void*)upconv8_kernel<K, 2, HAS_T, INBF>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipFuncSetAttribute((const void*)upconv8_kernel<K, 3, HAS_T, INBF>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipFuncSetAttribute((const void*)upconv8_kernel<K, 4, HAS_T, INBF>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    attr_set = true;
  }
  switch (a.N) {
    case 1: hipLaunchKernelGGL((upconv8_kernel<K, 1, HAS_T, INBF>), grid, block, lds, stream, a, tiles_per_n, tz, ty, tx); break;
    case 2: hipLaunchKernelGGL((upconv8_kernel<K, 2, HAS_T, INBF>), grid, block, lds, stream, a, tiles_per_n, tz, ty, tx); break;
    case 3: hipLaunchKernelGGL((upconv8_kernel<K, 3, HAS_T, INBF>), grid, block, lds, stream, a, tiles_per_n, tz, ty, tx); break;
    default: hipLaunchKernelGGL((upconv8_kernel<K, 4, HAS_T, INBF>), grid, block, lds, stream, a, tiles_per_n, tz, ty, tx); break;
  }
}

template <int K, bool HAS_T>
static void launch_upconv(const DArgs& a, int n, hipStream_t stream) {
  const dim3 grid(a.blocks_per_n, n), block(256);
  const size_t lds = ((size_t)4 * 65 * (K + 4) + 32) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)direct_upconv_kernel<K, 1, HAS_T>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    (void)hipFuncSetAttribute((const void*)direct_upconv_kernel<K, 2, HAS_T>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    (void)hipFuncSetAttribute((const void*)direct_upconv_kernel<K, 3, HAS_T>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    (void)hipFuncSetAttribute((const void*)direct_upconv_kernel<K, 4, HAS_T>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    attr_set = true;
  }
  switch (a.N) {
    case 1: hipLaunchKernelGGL((direct_upconv_kernel<K, 1, HAS_T>), grid, block, lds, stream, a); break;
    case 2: hipLaunchKernelGGL((direct_upconv_kernel<K, 2, HAS_T>), grid, block, lds, stream, a); break;
    case 3: hipLaunchKernelGGL((direct_upconv_kernel<K, 3, HAS_T>), grid, block, lds, stream, a); break;
    default: hipLaunchKernelGGL((direct_upconv_kernel<K, 4, HAS_T>), grid, block, lds, stream, a); break;
  }
}

template <int K, bool HAS_T>
static void launch_upconv_p(const DArgs& a, int n, bool bf, hipStream_t stream) {
  if (bf) launch_upconv_mfma<K, HAS_T>(a, n, stream); else launch_upconv<K, HAS_T>(a, n, stream);
}

template <bool HAS_T>
static void launch_row(const DArgs& a, int n, hipStream_t stream) {
  switch (a.K) {
    case 1: launch_row_n<1, HAS_T>(a, n, stream); break;
    case 2: launch_row_n<2, HAS_T>(a, n, stream); break;
    case 3: launch_row_n<3, HAS_T>(a, n, stream); break;
    default: launch_row_n<4, HAS_T>(a, n, stream); break;
  }
}

int direct_conv_run(const mmtta_conv_desc* d, const mmtta_tensor* x, const mmtta_norm_on_load* x_norm, const void* packed,
                    const float* bias, const mmtta_conv_epilogue* epi, const mmtta_tensor* y, int accumulate, float* stats,
                    const PSets& sets, hipStream_t stream) {
  DArgs a;
  a.ps = sets;
  a.in = tv(x); a.tin = nl(x_norm); a.out = tv(y);
  a.w = (const float*)packed; a.bias = bias;
  a.add = nullptr; a.asn = a.asd = a.ash = a.asw = 0; a.tadd = nl(nullptr);
  if (epi && epi->add) {
    const mmtta_tensor* ad = epi->add;
    MMTTA_CHECK(ad->ptr && is_cl(ad) && ad->n == y->n && ad->c == y->c && ad->d == y->d && ad->h == y->h && ad->w == y->w,
                MMTTA_ERR_INVALID, "conv: epilogue `add` must be channels-last with the shape of y");
    a.add = (const float*)ad->ptr; a.asn = ad->sn; a.asd = ad->sd; a.ash = ad->sh; a.asw = ad->sw;
    a.tadd = nl(&epi->add_norm);
  }
  a.K = x->c; a.N = y->c; a.ksize = d->ksize; a.stride = d->stride;
  a.transposed = (d->op == MMTTA_CONVT_FWD || d->op == MMTTA_CONV_DGRAD) ? 1 : 0;
  a.accumulate = accumulate;
  a.out_vec4 = (y->flags & MMTTA_TENSOR_OWNS_PAD) && y->sw == 4 && y->sc == 1 && y->sh % 4 == 0 && y->sd % 4 == 0 &&
               y->sn % 4 == 0 && ((uintptr_t)y->ptr) % 16 == 0;
  a.stats = stats; a.blocks_per_n = direct_blocks_per_n(d, x, y);
  const int T = d->ksize * d->ksize * d->ksize;
  const size_t lds = (size_t)T * a.K * 16 + (size_t)a.K * 8 + 32 * sizeof(float);
  static bool attr_set = false;
  if (!attr_set && lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)direct_conv_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void*)direct_conv_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    MMTTA_CHECK(e == hipSuccess, MMTTA_ERR_LAUNCH, "direct conv: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
    attr_set = true;
  }
  const bool has_t = a.tin.mean != nullptr || a.tin.scale != nullptr;
  const int variant = direct_variant(d, x);
  // bf16-stored operands: only the INPUT of the matrix-core up-convolution (the 64 / 32-channel concat buffer); the
  // <= 4-channel results and every other direct variant work on fp32-stored tensors
  // ... except the 4x4x4 matrix-tile convolution with ALL its thin tensors bf16-stored (8-byte voxels: the input-gradient
  // launch under bf16-stored thin gradients)
  const bool thin_bf = variant == 4 && is_bf16(x) && is_bf16(y) && (!(epi && epi->add) || is_bf16(epi->add)) &&
                       x->sw == 4 && y->sw == 4 && ((y->flags & MMTTA_TENSOR_OWNS_PAD) || y->c == 4) && ((uintptr_t)y->ptr) % 8 == 0 &&
                       (!(epi && epi->add) || epi->add->sw == 4);
  MMTTA_CHECK(thin_bf || (is_f32(y) && !(epi && epi->add && is_bf16(epi->add))), MMTTA_ERR_UNSUPPORTED, "direct conv: outputs are fp32-stored");
  if (variant == 5) {
    const int tz = (x->d + 3) / 4, ty = (x->h + 3) / 4, tx = (x->w + 7) / 8;
    if (is_bf16(x)) {
      if (a.K == 64) { if (has_t) launch_upconv8<64, true, true>(a, y->n, tz, ty, tx, stream); else launch_upconv8<64, false, true>(a, y->n, tz, ty, tx, stream); }
      else { if (has_t) launch_upconv8<32, true, true>(a, y->n, tz, ty, tx, stream); else launch_upconv8<32, false, true>(a, y->n, tz, ty, tx, stream); }
    } else {
      if (a.K == 64) { if (has_t) launch_upconv8<64, true, false>(a, y->n, tz, ty, tx, stream); else launch_upconv8<64, false, false>(a, y->n, tz, ty, tx, stream); }
      else { if (has_t) launch_upconv8<32, true, false>(a, y->n, tz, ty, tx, stream); else launch_upconv8<32, false, false>(a, y->n, tz, ty, tx, stream); }
    }
    return launch_status("up-convolution (2x2x2 gather GEMM)");
  }
  MMTTA_CHECK(is_f32(x) || (variant == 3 && d->dtype == MMTTA_BF16) || variant == 1 || thin_bf, MMTTA_ERR_UNSUPPORTED,
              "direct conv: a bf16-stored input is supported by the matrix-core up-convolution and the lanes-along-K kernel only");
  {  // 1x1x1 head on voxel-dense tensors, nothing fused: the streaming kernel
    auto dense = [](const mmtta_tensor* t) { return t->sc == 1 && t->sh == (int64_t)t->w * t->sw && t->sd == (int64_t)t->h * t->sh; };
    const bool head = d->op == MMTTA_CONV_FWD && d->ksize == 1 && (a.K == 32 || a.K == 64) && !has_t && stats == nullptr &&
                      !(epi && epi->add) && !accumulate && is_f32(y) && dense(x) && dense(y) && y->sw == 4 && a.out_vec4 &&
                      ((uintptr_t)x->ptr) % 16 == 0 && x->sw % (is_bf16(x) ? 8 : 4) == 0 && x->sn % (is_bf16(x) ? 8 : 4) == 0 &&
                      (long long)x->d * x->h * x->w * x->sw < (1LL << 31);
    if (head) {
      const long long dhw = (long long)y->d * y->h * y->w;
      long long blocks = (dhw + 255) / 256;
      if (blocks > 2048) blocks = 2048;
      const dim3 grid((unsigned)blocks, y->n), block(256);
      if (a.K == 32) { if (is_bf16(x)) hipLaunchKernelGGL((pointwise_head_kernel<32, true>), grid, block, 0, stream, a);
                       else hipLaunchKernelGGL((pointwise_head_kernel<32, false>), grid, block, 0, stream, a); }
      else { if (is_bf16(x)) hipLaunchKernelGGL((pointwise_head_kernel<64, true>), grid, block, 0, stream, a);
             else hipLaunchKernelGGL((pointwise_head_kernel<64, false>), grid, block, 0, stream, a); }
      return launch_status("1x1 head (streaming)");
    }
  }
  if (variant == 1) {
    const size_t kl_lds = (size_t)T * a.K * 16 + 32 * sizeof(float);
    if (a.K == 64) { if (has_t) launch_klane<16, true>(a, y->n, kl_lds, stream); else launch_klane<16, false>(a, y->n, kl_lds, stream); }
    else { if (has_t) launch_klane<8, true>(a, y->n, kl_lds, stream); else launch_klane<8, false>(a, y->n, kl_lds, stream); }
    return launch_status("direct conv (lanes along K)");
  }
  if (variant == 2) {
    if (has_t) launch_row<true>(a, y->n, stream); else launch_row<false>(a, y->n, stream);
    return launch_status("direct conv (row)");
  }
  if (variant == 4) {
    const dim3 grid(a.blocks_per_n, y->n), block(256);
    if (thin_bf) {
      if (g_thin_mfma == 3) {
        if (has_t) hipLaunchKernelGGL((conv3_mfma4_kernel<true, 2, true>), grid, block, 0, stream, a);
        else hipLaunchKernelGGL((conv3_mfma4_kernel<false, 2, true>), grid, block, 0, stream, a);
      } else if (g_thin_mfma == 2) {
        if (has_t) hipLaunchKernelGGL((conv3_mfma4_kernel<true, 4, true>), grid, block, 0, stream, a);
        else hipLaunchKernelGGL((conv3_mfma4_kernel<false, 4, true>), grid, block, 0, stream, a);
      } else {
        if (has_t) hipLaunchKernelGGL((conv3_mfma4_kernel<true, 8, true>), grid, block, 0, stream, a);
        else hipLaunchKernelGGL((conv3_mfma4_kernel<false, 8, true>), grid, block, 0, stream, a);
      }
      return launch_status("direct conv (4x4x4 matrix tiles, bf16-stored thin tensors)");
    }
    if (g_thin_mfma == 3) {
      if (has_t) hipLaunchKernelGGL((conv3_mfma4_kernel<true, 2>), grid, block, 0, stream, a);
      else hipLaunchKernelGGL((conv3_mfma4_kernel<false, 2>), grid, block, 0, stream, a);
    } else if (g_thin_mfma == 2) {
      if (has_t) hipLaunchKernelGGL((conv3_mfma4_kernel<true, 4>), grid, block, 0, stream, a);
      else hipLaunchKernelGGL((conv3_mfma4_kernel<false, 4>), grid, block, 0, stream, a);
    } else {
      if (has_t) hipLaunchKernelGGL((conv3_mfma4_kernel<true, 8>), grid, block, 0, stream, a);
      else hipLaunchKernelGGL((conv3_mfma4_kernel<false, 8>), grid, block, 0, stream, a);
    }
    return launch_status("direct conv (4x4x4 matrix tiles)");
  }
  if (variant == 3) {
    const bool bf = d->dtype == MMTTA_BF16;
    if (a.K == 64) { if (has_t) launch_upconv_p<64, true>(a, y->n, bf, stream); else launch_upconv_p<64, false>(a, y->n, bf, stream); }
    else { if (has_t) launch_upconv_p<32, true>(a, y->n, bf, stream); else launch_upconv_p<32, false>(a, y->n, bf, stream); }
    return launch_status("direct up-convolution");
  }
  if (has_t) hipLaunchKernelGGL(direct_conv_kernel<true>, dim3(a.blocks_per_n, y->n), dim3(256), lds, stream, a);
  else hipLaunchKernelGGL(direct_conv_kernel<false>, dim3(a.blocks_per_n, y->n), dim3(256), lds, stream, a);
  return launch_status("direct conv");
}

}  // namespace mmtta
