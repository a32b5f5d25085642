// Implicit-GEMM convolution on the CDNA4 matrix cores (gfx950), channels-last activations.
//
// One kernel family serves Conv3d forward, its input gradient, ConvTranspose3d forward and its
// input gradient (reference call sites: src/models/unet.py:68-69 via monai UNet,
// src/models/unet_multimodal_midfusion.py:204-267; backward from
// src/core/trainers/seg_trainer.py:142).  All four are the same "gather GEMM":
//
//     out[g*so + o][n] = sum_{tap} sum_{k} T(in[g*si + d_tap][k]) * Wp[slab_tap][k][n]
//
// g runs over an output sub-grid.  Conv: so=1, si=stride, d = k-pad.  Input gradient of a stride-1
// conv: d = pad-k.  The stride-2 transposed forms are split into 8 output parity classes (so=2),
// each a small dense conv over 1..8 taps - no zero-stuffing, no atomics, no wasted MACs.
//
// Work decomposition (wave = 64 lanes, 4 waves per workgroup, one workgroup = one spatial box):
//   * the input box of the tile (tile + halo, KCI channels at a time) is staged ONCE in LDS and
//     re-used by every tap (27x re-use for k=3) - global reads are ~1.5-2.4x the input instead
//     of 27x; norm+ReLU of the producer is applied while staging (norm on load), padding voxels
//     are written as exact zeros;
//   * each wave owns MB 32-row blocks x one 32-column block: v_mfma_f32_32x32x2_f32 (exact fp32,
//     one operand element per lane, so the LDS image needs no fragment layout) with the
//     weight fragment fetched straight from the packed image (L1/L2 resident) one tap ahead;
//   * LDS voxel stride KCI+1 words: the 32 rows of an A fragment fall on distinct banks;
//   * epilogue: bias, optional fused residual add, store (128-B segments), per-tile
//     sum / sum-of-squares for the following InstanceNorm (deterministic partial slab, no atomics);
//   * small spatial extents (8^3 bottleneck) get split-K over channel blocks through a workspace
//     so that >= 256 workgroups exist.
#include "common.h"

namespace mmtta {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
  bf16x2 v;
  v[0] = (__bf16)lo;   // plain casts: hipcc emits v_cvt_pk_bf16_f32 (round to nearest even, NaN preserved)
  v[1] = (__bf16)hi;
  return __builtin_bit_cast(unsigned int, v);
}

struct Taps {      // host-side description of one tap set
  int n;
  int slab[27];
  signed char dz[27], dy[27], dx[27];
  int zmin, ymin, xmin;
  int zext, yext, xext;
};

// One output parity class of a launch (a plain conv is a single class with all its taps).  The stride-2
// transposed forms run their 8 classes in ONE launch, interleaved with the tile index (class_of_tile below).
struct ClassInfo {
  int tap0, ntaps;
  int oz, oy, ox;
  int zmin, ymin, xmin;
  int zext, yext, xext;
  int Dg, Hg, Wg;
  // box extents of this class for the launch's tile, with reciprocals for exact division of a box index < 2^16:
  // q = __umulhi(v, ceil(2^32 / d))
  int BX, BXY;
  unsigned mBX, mBXY;
};

struct GArgs {
  const float* in; long long isn, isd, ish, isw; int Ci, Di, Hi, Wi;
  NL tin;
  float* out; long long osn, osd, osh, osw; int Co, Do, Ho, Wo;
  int so;
  int si;
  const float* wp; int Kp, Np;
  const float* bias;
  PSets ps;        // per-item parameter sets: wp / bias are set 0's (common.h)
  const float* add; long long asn, asd, ash, asw; NL tadd;
  int accumulate;
  float* stats; int stats_rows_per_n;
  float* ws; int ksplit, stages_per_split, nstages;
  int tz, ty, tx;
  int in_bf, out_bf, add_bf;   // storage of the three activation operands: 1 = bf16 elements (common.h storage helpers)
  int vec4;
  int ovec;        // 16-byte epilogue stores (out / add rows 16-byte aligned, Co % 4 == 0)
  int coef_off;    // word offset of the coefficient table behind the LDS box image
  int rowload;     // the input admits the row-structured loader (alignment, 24-bit strides, < 2^31 elements)
  int flip27;      // canonical 27-tap stride-1 stage (rowload): tap t reads weight slab 26 - t (stride-1 input gradient) instead of t
  int ncls;
  ClassInfo cls[8];
  int toff[27];   // box-relative voxel offset of each tap (int32 tables: read with SCALAR loads)
  int slab[27];   // weight slab of each tap
};

// Launch tile index -> (parity class, tile inside the class).  The classes of one tile are NEIGHBOURS in the index, the
// 8-tap class first: the 8 XCDs each run one contiguous eighth of the index (xcd_contiguous_id), and with the classes laid
// out one after the other XCD c ran class c alone - 1 tap's work on XCD 0 against 8 taps' on XCD 7, the launch as slow as
// its heaviest eighth (2.4x the mean of 27/8 taps).  Interleaved, every XCD carries the same mix and the 8 classes of a tile
// read their shared input box from one L2.
__device__ __forceinline__ void class_of_tile(const GArgs& a, int tile, int& cidx, int& t) {
  if (a.ncls == 8) { cidx = 7 - (tile & 7); t = tile >> 3; }
  else { cidx = 0; t = tile; }
}

// MFMA row index v of a tile -> local voxel.  PERM (bf16 images): inside every 32-row block (4 y-rows x 8 x) the
// rows are permuted so that the 16 lanes ds_read_b128 services together ({0-3,12-15,20-27} and {4-11,16-19,28-31})
// hold the voxels of y-rows {0,2} and {1,3}: with an LDS x-row pitch of LP = 12 voxels of 48 bytes their
// 16-byte slots are all distinct modulo 16 - the A-fragment reads are bank-conflict free (3-way before).
template <int TZ, int TY, int TX, bool PERM>
__device__ __forceinline__ void row_to_local(int v, int& zl, int& yl, int& xl) {
  if (PERM) {
    static_assert(TX == 8 && TY % 4 == 0, "the permutation assumes 4 x 8 row blocks");
    const int r = v & 31, q = r >> 2;
    const int y = (0xEB14 >> (2 * q)) & 3;           // q: 0 1 2 3 4 5 6 7 -> y: 0 1 1 0 3 2 2 3
    const int xh = (q >> 1) & 1;
    v = (v & ~31) + y * 8 + xh * 4 + (r & 3);
  }
  xl = v % TX;
  yl = (v / TX) % TY;
  zl = v / (TX * TY);
}

constexpr int LDS_PITCH_BF16 = 12;   // x-row pitch (voxels) of the bf16 LDS image of stride-1 gathers

// Box-relative voxel offset of tap t of the canonical 3x3x3 stride-1 stage (taps in ascending (z, y, x) box order, RBY box
// rows per z plane): a compile-time constant once the tap loops are unrolled, i.e. an immediate of the ds_read.
__host__ __device__ constexpr int tap27_off(int t, int rby) { return ((t / 9) * rby + (t / 3) % 3) * LDS_PITCH_BF16 + t % 3; }

// Epilogue with 16-byte stores.  An MFMA accumulator block holds ONE output channel per lane (32 lanes = a 128-byte
// voxel row) and 16 voxels in its registers, so storing straight from it takes 16 four-byte-per-lane stores per block -
// measured in the producer/consumer kernel (s_memtime): 42k cycles per 512-voxel tile, more than its four MFMA stages
// together; the store path is issue-bound at ~1.6 bytes per cycle and CU that way.  Here every 32 x 32 block goes
// through a wave-private LDS tile and comes back as float4 = 4 channels of one voxel per lane: 4 stores of 1 KiB per
// block (8 lanes cover a voxel row), and the fused add / accumulate operands are fetched 16 bytes per lane too.
// `tr`: 32 x 36 floats of this wave.  Per-lane sums for the following norm: channels colbase + 4*(lane&7) + 0..3,
// already added up over the 8 row lanes (valid in lanes 0-7).
template <int TZ, int TY, int TX, int MB, bool PERM, bool ABF>
__device__ __forceinline__ void epilogue_vec16(const GArgs& a, const float* biasp, const ClassInfo& ci, f32x16 (&acc)[MB], float* tr,
                                               int lane, int rowblock0, int colbase, bool colact, int n, int gz0, int gy0, int gx0,
                                               float (&ssum)[4], float (&ssq)[4]) {
  // kernel-argument fields used per row: opaque scalar copies (see the tap tables of the producer/consumer kernel)
  int osd = (int)a.osd, osh = (int)a.osh, osw = (int)a.osw, asd = (int)a.asd, ash_ = (int)a.ash, asw = (int)a.asw;
  int so = a.so, Do = a.Do, Ho = a.Ho, Wo = a.Wo, coz = ci.oz, coy = ci.oy, cox = ci.ox, cDg = ci.Dg, cHg = ci.Hg, cWg = ci.Wg;
  asm volatile("" : "+s"(osd), "+s"(osh), "+s"(osw), "+s"(asd), "+s"(ash_), "+s"(asw), "+s"(so), "+s"(Do), "+s"(Ho), "+s"(Wo));
  asm volatile("" : "+s"(coz), "+s"(coy), "+s"(cox), "+s"(cDg), "+s"(cHg), "+s"(cWg));
  const int h = lane >> 5, r = lane & 31;
  const int rsub = lane >> 3, c4 = (lane & 7) * 4;
  const int colv = colbase + c4;
  const bool cvok = colact && colv < a.Co;                 // Co % 4 == 0 on this path: all four channels or none
  const int colc = min(colv, a.Co - 4);
  float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float asc[4] = {1.f, 1.f, 1.f, 1.f}, ash[4] = {0.f, 0.f, 0.f, 0.f};
  if (biasp) bias4 = *reinterpret_cast<const float4*>(biasp + colc);
  if (a.add) nl_coeff_vec<4>(a.tadd, n, a.Co, colc, asc, ash);
#pragma unroll
  for (int j = 0; j < 4; ++j) { ssum[j] = 0.f; ssq[j] = 0.f; }
  const float* addb = a.add;
  float* outb = a.out;
  const long long abase = (long long)n * a.asn + colc, obase = (long long)n * a.osn + colc;
  // value path of one (block, row slot): bias, fused add, accumulate, store, statistics - shared by both address paths
  auto finish = [&](const float4& val, const float4& addq, const float4& oldq, bool okk, int oo) {
    float v[4] = {val.x + bias4.x, val.y + bias4.y, val.z + bias4.z, val.w + bias4.w};
    if (a.add) {
      const float av[4] = {addq.x, addq.y, addq.z, addq.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] += nl_apply(av[j], asc[j], ash[j], a.tadd.relu);
    }
    if (a.accumulate) { v[0] += oldq.x; v[1] += oldq.y; v[2] += oldq.z; v[3] += oldq.w; }
    if (okk) {
      st4_t<ABF>(outb, obase + oo, make_float4(v[0], v[1], v[2], v[3]));
#pragma unroll
      for (int j = 0; j < 4; ++j) { ssum[j] += v[j]; ssq[j] += v[j] * v[j]; }
    }
  };
  // Interior tiles of the un-decomposed ops (every row inside the tensor, output voxel = grid voxel): the offsets of the
  // 4 row slots of a block are wave-uniform terms plus two per-lane constants (the row permutation keeps x = 4 (k & 1) +
  // (lane >> 3 & 3) and picks y from one of two compile-time values by lane bit 5), ~2 vector instructions per row
  // instead of ~35 (decode, bounds, clamps, six 32-bit multiplies); and because nothing depends on the accumulators, the
  // fused-add (or accumulate) operand of ALL blocks is requested up front - one exposed round trip per tile, not one per
  // block (the weight-fragment registers are free by now).
  const bool interior = PERM && so == 1 && coz == 0 && coy == 0 && cox == 0 && gz0 + TZ <= min(cDg, Do) && gy0 + TY <= min(cHg, Ho) &&
                        gx0 + TX <= min(cWg, Wo);
  if (interior) {
    const bool lb = (rsub >> 2) != 0;
    const int lxo = (rsub & 3) * osw, lxa = (rsub & 3) * asw;
    constexpr int T = 0xEB14;                         // row_to_local's y table
    auto row_off = [&](int mb, int k, int sd_, int sh_, int sw_, int lx) {     // (block, row slot) -> element offset
      const int rb = rowblock0 + mb;
      const int zl = TY == 4 ? rb : (rb >> 1), yb = TY == 4 ? 0 : (rb & 1) * 4;
      const int u = (gz0 + zl) * sd_ + (gy0 + yb) * sh_ + (gx0 + 4 * (k & 1)) * sw_;              // wave-uniform
      const int y0 = (T >> (2 * (2 * k))) & 3, y1 = (T >> (2 * (2 * k + 1))) & 3;
      return (lb ? u + y1 * sh_ : u + y0 * sh_) + lx;
    };
    const bool one_operand = (a.add != nullptr) != (a.accumulate != 0);
    float4 pre[MB][4];
    if (one_operand) {
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int k = 0; k < 4; ++k)
          pre[mb][k] = a.add ? ld4_t<ABF>(addb, abase + row_off(mb, k, asd, ash_, asw, lxa))
                             : ld4_t<ABF>(outb, obase + row_off(mb, k, osd, osh, osw, lxo));
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
      for (int i = 0; i < 16; ++i) tr[((i & 3) + 8 * (i >> 2) + 4 * h) * 36 + r] = acc[mb][i];
      float4 val[4], addv[4], oldv[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) val[k] = *reinterpret_cast<const float4*>(tr + (rsub + 8 * k) * 36 + c4);
      if (!one_operand) {
        if (a.add) {
#pragma unroll
          for (int k = 0; k < 4; ++k) addv[k] = ld4_t<ABF>(addb, abase + row_off(mb, k, asd, ash_, asw, lxa));
        }
        if (a.accumulate) {
#pragma unroll
          for (int k = 0; k < 4; ++k) oldv[k] = ld4_t<ABF>(outb, obase + row_off(mb, k, osd, osh, osw, lxo));
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int oo = row_off(mb, k, osd, osh, osw, lxo);
        if (one_operand) finish(val[k], pre[mb][k], pre[mb][k], cvok, oo);
        else finish(val[k], addv[k], oldv[k], cvok, oo);
      }
    }
  } else {
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
    for (int i = 0; i < 16; ++i) tr[((i & 3) + 8 * (i >> 2) + 4 * h) * 36 + r] = acc[mb][i];
    int ooff[4], aoff[4];
    bool ok[4];
    float4 val[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int row = rsub + 8 * k;
      val[k] = *reinterpret_cast<const float4*>(tr + row * 36 + c4);
      int zl, yl, xl;
      row_to_local<TZ, TY, TX, PERM>((rowblock0 + mb) * 32 + row, zl, yl, xl);
      const int gz = gz0 + zl, gy = gy0 + yl, gx = gx0 + xl;
      const int oz = gz * so + coz, oy = gy * so + coy, ox = gx * so + cox;
      ok[k] = cvok && gz < cDg && gy < cHg && gx < cWg && oz < Do && oy < Ho && ox < Wo;
      const int cz = min(oz, Do - 1), cy = min(oy, Ho - 1), cx = min(ox, Wo - 1);
      ooff[k] = cz * osd + cy * osh + cx * osw;
      aoff[k] = cz * asd + cy * ash_ + cx * asw;
    }
    float4 addv[4], oldv[4];
    if (a.add) {
#pragma unroll
      for (int k = 0; k < 4; ++k) addv[k] = ld4_t<ABF>(addb, abase + aoff[k]);
    }
    if (a.accumulate) {
#pragma unroll
      for (int k = 0; k < 4; ++k) oldv[k] = ld4_t<ABF>(outb, obase + ooff[k]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) finish(val[k], addv[k], oldv[k], ok[k], ooff[k]);
  }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) { ssum[j] += __shfl_xor(ssum[j], o, 64); ssq[j] += __shfl_xor(ssq[j], o, 64); }
  }
}

constexpr int EPI_TILE_FLOATS = 32 * 36;     // LDS words of one wave's transposition tile

// BF == false: fp32 operands, v_mfma_f32_32x32x2_f32 (bit-exact fp32 products, the parity path).
// BF == true : activations are rounded to bf16 while they are staged, weights come from a bf16 image,
//              v_mfma_f32_32x32x16_bf16 with fp32 accumulation (16x the matrix rate: these layers turn
//              HBM/LDS bound).  Storage, statistics, epilogue stay fp32 in both modes.
// OCC = workgroups the register budget admits per CU (launch bound in waves per SIMD): 2 for the four-block tiles
// (<= 256 registers), 4 for the lean two-block tile <1,2,4,8,8,16> of the 32-channel 64^3 layers (<= 128 registers,
// smaller weight-fragment groups, no box prefetch): twice the resident waves to cover each other's load, LDS and
// store latencies (MMTTA_OPT_IGEMM_LEAN).
// ABF: the three activation operands (input, output, fused add) are bf16-STORED (forward convolutions of bf16 precision);
// false: all fp32 (every input-gradient launch, fp32 precision).  Compile-time: see MMTTA_BF_DISPATCH in common.h.
template <int NB, int MB, int TZ, int TY, int TX, int KCI, bool BF, int OCC = 2, bool ABF = false>
__global__ __launch_bounds__(256, OCC) void igemm_kernel(GArgs a) {
  extern __shared__ float lds[];
  // LDS voxel stride: fp32: KCI+1 words (odd: the 32 rows of a fragment hit distinct banks);
  // bf16: KCI+8 halfwords (16-byte slots stay aligned for ds_read_b128)
  constexpr int VS = BF ? (KCI + 8) : (KCI + 1);
  static_assert(!BF || KCI % 16 == 0, "bf16 stages are multiples of the MFMA K=16");
  constexpr int MT = TZ * TY * TX;
  constexpr int MG = 4 / NB;
  static_assert(MT == 32 * MB * MG, "tile rows must equal 32*MB*(4/NB)");

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cb = wave % NB, mg = wave / NB;
  const int h = lane >> 5, r = lane & 31;

  // workgroups are dealt round-robin over the 8 XCDs: give each XCD one CONTIGUOUS run of tiles (a z-slab of the
  // volume), so that the halo rows neighbouring tiles share are served by that XCD's L2 instead of being fetched again
  // (the whole 3-D grid is renumbered, tile index fastest: the tiles that stream the SAME weight panel - same column
  // group, same K split - then also share an XCD, which matters for the weight-bound 8^3 / 16^3 levels)
  const unsigned lflat = xcd_contiguous_id(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z),
                                           gridDim.x * gridDim.y * gridDim.z);
  const int bx = (int)(lflat % gridDim.x);
  const int lby = (int)((lflat / gridDim.x) % gridDim.y), lbz = (int)(lflat / (gridDim.x * gridDim.y));
  int cidx, t;
  class_of_tile(a, bx, cidx, t);
  const ClassInfo ci = a.cls[cidx];
  const int tile_in_n = t % (a.tz * a.ty * a.tx);
  const int txi = t % a.tx; t /= a.tx;
  const int tyi = t % a.ty; t /= a.ty;
  const int tzi = t % a.tz;
  const int n = t / a.tz;
  const float* wpn = pset_packed(a.ps, a.wp, n);        // this batch item's parameter set (workgroup-uniform)
  const float* biasn = pset_bias(a.ps, a.bias, n);
  const int gz0 = tzi * TZ, gy0 = tyi * TY, gx0 = txi * TX;
  const int BZ = (TZ - 1) * a.si + ci.zext + 1;
  const int BY = (TY - 1) * a.si + ci.yext + 1;
  const int BX = (TX - 1) * a.si + ci.xext + 1;
  const int boxvox = BZ * BY * BX;
  const int LP = (BF && a.si == 1) ? LDS_PITCH_BF16 : BX;      // LDS x-row pitch in voxels
  const int iz0 = gz0 * a.si + ci.zmin, iy0 = gy0 * a.si + ci.ymin, ix0 = gx0 * a.si + ci.xmin;

  const int colbase = (lby * NB + cb) * 32;
  const bool colact = colbase < a.Np;

  int rowaddr[MB];   // LDS word address of (row's voxel, channel h) inside the box
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    int zl, yl, xl;
    row_to_local<TZ, TY, TX, BF>((mg * MB + mb) * 32 + r, zl, yl, xl);
    rowaddr[mb] = (((zl * a.si) * BY + yl * a.si) * LP + xl * a.si) * VS + (BF ? 8 * h : h);
  }

  f32x16 acc[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[mb][i] = 0.f;

  const int ks0 = lbz * a.stages_per_split;
  const int ks1 = min(a.nstages, ks0 + a.stages_per_split);
  // batch item base: element offsets are applied by the storage helpers (the pointer itself is never advanced, so the
  // same code addresses 2-byte and 4-byte elements)
  const float* inb = a.in;
  const long long inoff = (long long)n * a.isn;

  // Row-structured loader of the bf16 3x3x3 stride-1 stages (the bulk of the network).  With several volumes in flight
  // the chip is bound by vector-ALU issue (profiles/r02c_sq_counters.md) and the generic walk below spends ~100 vector
  // instructions per 8-channel item on index decoding, bounds and 64-bit addressing.  Here an item's geometry is paid
  // ONCE PER TILE: a thread owns one (x, channel chunk) column of the box and walks the box rows RPP at a time, so its
  // 32-bit element offset of pass P (voff[P], channel base folded into the wave-uniform pointer) and its in-bounds bit
  // are the same for every stage, the LDS address is a per-thread constant plus an immediate, and a stage costs the
  // conversion / norm-on-load / pack only.  The first PG passes of stage ks+1 are requested during the MFMA phase of
  // stage ks - after its last weight-fragment request, because loads return in order - and land under it.
  constexpr int RCV8 = BF ? KCI / 8 : 1;                  // 8-channel chunks per voxel
  constexpr int RBY = TY + 2, RBZ = TZ + 2, RBX = TX + 2;
  constexpr int RIPR = RBX * RCV8;                        // items per box row
  constexpr int RRPP = 256 / RIPR;                        // box rows per pass (threads beyond repeat the last row)
  constexpr int RNROW = RBZ * RBY;
  constexpr int RGP = (RNROW + RRPP - 1) / RRPP;          // passes per stage
  constexpr int RPG = !BF ? 0 : (OCC > 2 ? 0 : (KCI == 16 ? (ABF ? 4 : 2) : 0));   // of which prefetched across the MFMA phase
  constexpr int RGPA = BF ? RGP : 1;
  const bool fast = BF && a.rowload && ci.ntaps == 27 && a.si == 1;      // workgroup-uniform: all 256 threads stage
  const bool pipe = fast && RPG > 0;
  Oct8<ABF> gv[RGPA];
  unsigned voff[RGPA];
  unsigned pok = 0u;
  // the 16-channel stages keep the offsets across the stages; the 32-channel stages have no registers to spare during
  // their MFMA phase (two weight-fragment sets of 40) and rebuild them per stage (~12 instructions per item, still 1/8 of
  // the generic walk) - `rrsub` goes through an opaque copy there, or the compiler hoists the rebuild out of the K loop
  constexpr bool RHOIST = KCI == 16;
  const int rrsub = min(tid / RIPR, RRPP - 1), rrem = tid % RIPR, rbx = rrem / RCV8, rcv = rrem % RCV8;
  auto geometry = [&]() {
    int rs = rrsub;
    if constexpr (!RHOIST) asm volatile("" : "+v"(rs));
    const int ix = ix0 + rbx;
    const bool xok = (unsigned)ix < (unsigned)a.Wi;
    const unsigned xoff = __umul24((unsigned)min(max(ix, 0), a.Wi - 1), (unsigned)a.isw) + rcv * 8;
    pok = 0u;
    static_for<0, RGP>([&](auto pc) {
      constexpr int P = decltype(pc)::value;
      const int row = min(RRPP * P + rs, RNROW - 1);
      const int bz = row / RBY, by = row - bz * RBY;
      const int iz = iz0 + bz, iy = iy0 + by;
      const bool ok = xok && (unsigned)iz < (unsigned)a.Di && (unsigned)iy < (unsigned)a.Hi;
      pok |= (ok ? 1u : 0u) << P;
      voff[P] = __umul24((unsigned)min(max(iz, 0), a.Di - 1), (unsigned)a.isd) + __umul24((unsigned)min(max(iy, 0), a.Hi - 1), (unsigned)a.ish) + xoff;
    });
  };
  unsigned short* lh0 = reinterpret_cast<unsigned short*>(lds);
  unsigned short* rst = lh0 + (rrsub * LDS_PITCH_BF16 + rbx) * VS + rcv * 8;       // + P * RRPP * LDS_PITCH_BF16 * VS
  // norm-on-load coefficients of every channel this workgroup will stage, once, in LDS behind the box image (a
  // per-stage fetch from global memory would be an exposed round trip in front of every commit)
  float* coef = lds + a.coef_off;
  const int cbase = ks0 * KCI, nch = (ks1 - ks0) * KCI;
  if (fast) {
    for (int cch = tid; cch < nch; cch += 256) {
      float sc1 = 0.f, sh1 = 0.f;
      if (cbase + cch < a.Ci) nl_coeff(a.tin, n, a.Ci, cbase + cch, sc1, sh1);     // channels past the last one stage as zeros
      coef[cch] = sc1;
      coef[nch + cch] = sh1;
    }
    if constexpr (RHOIST) geometry();
  }
  // channel part of an item's address: wave-uniform (folded into the pointer); the last stage of a Ci that is no multiple
  // of KCI clamps to the last valid chunk (its coefficients are zero)
  auto stage_base = [&](int ks) {
    return reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.in) + ((long long)n * a.isn + (long long)ks * KCI) * (ABF ? 2 : 4));
  };
  const int cmax8 = ABF ? ((a.Ci - 1) & ~7) : ((a.Ci - 1) & ~3);
  auto load_item = [&](auto pc, const float* sb, int c0) {
    constexpr int P = decltype(pc)::value;
    // chunks past the tensor's last one (partial last stage) re-read the last valid chunk; their coefficients are zero
    const int over = max(c0 + rcv * 8 - cmax8, 0), over_hi = max(c0 + rcv * 8 + 4 - cmax8, 0);
    gv[P] = oct8_ld<ABF>(sb, voff[P] - over, voff[P] + 4 - over_hi);
  };
  auto issue = [&](int ks) {
    const float* sb = stage_base(ks);
    static_for<0, RPG>([&](auto pc) { load_item(pc, sb, ks * KCI); });
  };
  if (fast && ks0 < ks1) {
    if (pipe) issue(ks0);
    __syncthreads();      // coefficients visible
  }

  // Weight fragment of (tap t, 16-channel step k2) of the canonical stage: a wave-uniform 64-bit base (scalar registers:
  // parameter set, stage, slab - the slab walks up for the forward form, down for the mirrored input gradient) plus ONE
  // 32-bit per-lane byte offset, so a request costs two scalar adds instead of a 64-bit multiply-add per lane
  // (measured on the 64^3 32-channel layers: 7 scalar + 1 vector instruction per fragment before).
  // (compiled into the lean 64^3 tile only: beside the table-driven stage it costs the four-block tiles, which sit at the
  // 256-register limit, 200-450 bytes of spills)
  constexpr bool CANON_CFG = BF && OCC > 2;
  const long long wslab16 = (long long)(a.Kp / 8) * a.Np * 16;                    // bytes per tap slab
  const char* const wuni = reinterpret_cast<const char*>(wpn) + (a.flip27 ? 26 * wslab16 : 0);
  const long long wstep16 = a.flip27 ? -wslab16 : wslab16;
  const unsigned wlane16 = (unsigned)(h * a.Np + colbase + r) * 16u;
  auto wfrag27 = [&](int c0, int t, int k2) {
    const char* ub = wuni + ((long long)(c0 / 8) * a.Np + (long long)k2 * 2 * a.Np) * 16 + t * wstep16;
    return *reinterpret_cast<const uint4*>(ub + wlane16);
  };

  for (int ks = ks0; ks < ks1; ++ks) {
    const int stage = ks;
    const int c0 = ks * KCI;
    // the first group of weight fragments of this stage is requested before the staging pass, so its L2 round trip
    // hides behind the box loads instead of opening the MFMA phase (full 27-tap stages of the bf16 path only)
    constexpr int WG0 = BF ? (OCC > 2 ? 3 : (KCI == 16 ? 9 : 5)) : 1;
    constexpr int WKS = BF ? KCI / 16 : 1;
    uint4 wfirst[WG0][WKS];
    bool wfirst_ok = false;
    if constexpr (BF) {
      if (colact && ci.ntaps == 27 && ((min(KCI, a.Ci - c0) + 15) >> 4) == WKS) {   // same test as the MFMA section
        wfirst_ok = true;
        const long long slabsz8 = (long long)(a.Kp / 8) * a.Np;
        if (CANON_CFG && fast) {
#pragma unroll
          for (int t = 0; t < WG0; ++t)
#pragma unroll
            for (int k2 = 0; k2 < WKS; ++k2) wfirst[t][k2] = wfrag27(c0, t, k2);
        } else {
          const uint4* wq0 = reinterpret_cast<const uint4*>(wpn) + (long long)(c0 / 8 + h) * a.Np + colbase + r;
#pragma unroll
          for (int t = 0; t < WG0; ++t)
#pragma unroll
            for (int k2 = 0; k2 < WKS; ++k2) wfirst[t][k2] = wq0[(a.slab + ci.tap0)[t] * slabsz8 + k2 * 2 * a.Np];
        }
      }
    }
    // ---------------- stage the input box (KCI channels) into LDS ----------------
    if constexpr (BF) {
      unsigned short* lh = reinterpret_cast<unsigned short*>(lds);
      if (fast) {
        const float* sb = stage_base(ks);
        // what the prefetch had no registers for comes in rounds of RB passes (all loads of a round in flight together);
        // the first round is requested before the prefetched passes are committed
        constexpr int RB = ABF ? 6 : 4;
        constexpr int R1 = RPG + RB < RGP ? RPG + RB : RGP;
        if constexpr (!RHOIST) geometry();
        static_for<RPG, R1>([&](auto pc) { load_item(pc, sb, c0); });
        float sc[8], sh[8];
        {
          const float4* cq = reinterpret_cast<const float4*>(coef + (c0 - cbase) + rcv * 8);
          const float4* hq = reinterpret_cast<const float4*>(coef + nch + (c0 - cbase) + rcv * 8);
          const float4 s0 = cq[0], s1 = cq[1], h0 = hq[0], h1 = hq[1];
          sc[0] = s0.x; sc[1] = s0.y; sc[2] = s0.z; sc[3] = s0.w; sc[4] = s1.x; sc[5] = s1.y; sc[6] = s1.z; sc[7] = s1.w;
          sh[0] = h0.x; sh[1] = h0.y; sh[2] = h0.z; sh[3] = h0.w; sh[4] = h1.x; sh[5] = h1.y; sh[6] = h1.z; sh[7] = h1.w;
        }
        const float relu_lo = a.tin.relu ? 0.f : -__builtin_inff();
        auto commit_item = [&](auto pc) {
          constexpr int P = decltype(pc)::value;
          float v[8];
          oct8_f8(gv[P], v);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = fmaxf(fmaf(v[j], sc[j], sh[j]), relu_lo);
          const unsigned okm = ((pok >> P) & 1u) ? 0xffffffffu : 0u;
          uint4 pk;
          pk.x = pack_bf16x2(v[0], v[1]) & okm; pk.y = pack_bf16x2(v[2], v[3]) & okm;
          pk.z = pack_bf16x2(v[4], v[5]) & okm; pk.w = pack_bf16x2(v[6], v[7]) & okm;
          if (RNROW % RRPP == 0 || P + 1 < RGP || RRPP * P + rrsub < RNROW)       // the last pass may be partial
            *reinterpret_cast<uint4*>(rst + P * (RRPP * LDS_PITCH_BF16 * VS)) = pk;
        };
        static_for<0, R1>(commit_item);
        static_for_step<R1, RGP, RB>([&](auto r0) {
          constexpr int A = decltype(r0)::value, B = A + RB < RGP ? A + RB : RGP;
          static_for<A, B>([&](auto pc) { load_item(pc, sb, c0); });
          static_for<A, B>(commit_item);
        });
      } else if (a.vec4) {
        constexpr int CV8 = KCI / 8;
        const int cv = tid % CV8;          // 256 % CV8 == 0
        const int c = c0 + cv * 8;
        float sc[8], sh[8];
        constexpr int STEP = 256 / CV8;
        const int first_item = 0;
        nl_coeff_vec<8>(a.tin, n, a.Ci, c, sc, sh);
        // U items per trip: all their global loads are issued before the first use (one exposed latency per
        // trip instead of one per item)
        // all of a thread's items in as few trips as the register budget allows: one exposed memory latency per trip
        constexpr int U = OCC > 2 ? 2 : 4;
        const bool tail = (a.Ci & 7) != 0;       // only then can lanes beyond Ci hold uninitialised padding
        for (int bv0 = tid / CV8 + first_item * STEP; bv0 < boxvox; bv0 += U * STEP) {
          float4 x0[U], x1[U];
          bool ok[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int bv = bv0 + u * STEP;
            const int bz = (int)__umulhi((unsigned)bv, ci.mBXY), brem = bv - bz * ci.BXY;
            const int by = (int)__umulhi((unsigned)brem, ci.mBX), bx = brem - by * ci.BX;
            const int iz = iz0 + bz, iy = iy0 + by, ix = ix0 + bx;
            ok[u] = bv < boxvox && (unsigned)iz < (unsigned)a.Di && (unsigned)iy < (unsigned)a.Hi &&
                    (unsigned)ix < (unsigned)a.Wi && c < a.Ci;
            x0[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            x1[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok[u]) {
              const long long so = inoff + (long long)iz * a.isd + (long long)iy * a.ish + (long long)ix * a.isw + c;
              if constexpr (ABF) ld8_t<true>(inb, so, x0[u], x1[u]);      // rows of bf16 tensors are padded to 8 channels
              else {
                x0[u] = *reinterpret_cast<const float4*>(inb + so);
                if (c + 4 < a.Ci) x1[u] = *reinterpret_cast<const float4*>(inb + so + 4);
              }
            }
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int bv = bv0 + u * STEP;
            if (bv < boxvox) {
              uint4 pk = make_uint4(0u, 0u, 0u, 0u);
              if (ok[u]) {
                const float xs[8] = {x0[u].x, x0[u].y, x0[u].z, x0[u].w, x1[u].x, x1[u].y, x1[u].z, x1[u].w};
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                  v[j] = nl_apply(xs[j], sc[j], sh[j], a.tin.relu);
                  if (tail && c + j >= a.Ci) v[j] = 0.f;
                }
                pk.x = pack_bf16x2(v[0], v[1]); pk.y = pack_bf16x2(v[2], v[3]);
                pk.z = pack_bf16x2(v[4], v[5]); pk.w = pack_bf16x2(v[6], v[7]);
              }
              const int bz = (int)__umulhi((unsigned)bv, ci.mBXY), brem = bv - bz * ci.BXY;
              const int by = (int)__umulhi((unsigned)brem, ci.mBX), bx = brem - by * ci.BX;
              *reinterpret_cast<uint4*>(lh + ((bz * BY + by) * LP + bx) * VS + cv * 8) = pk;
            }
          }
        }
      } else {
        const int cc = tid % KCI;
        const int c = c0 + cc;
        float sc, sh;
        nl_coeff_vec<1>(a.tin, n, a.Ci, c, &sc, &sh);
        for (int bv = tid / KCI; bv < boxvox; bv += 256 / KCI) {
          const int bz = (int)__umulhi((unsigned)bv, ci.mBXY), brem = bv - bz * ci.BXY;
          const int by = (int)__umulhi((unsigned)brem, ci.mBX), bx = brem - by * ci.BX;
          const int iz = iz0 + bz, iy = iy0 + by, ix = ix0 + bx;
          float v = 0.f;
          if ((unsigned)iz < (unsigned)a.Di && (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi &&
              c < a.Ci)
            v = nl_apply(ld1_t<ABF>(inb, inoff + (long long)iz * a.isd + (long long)iy * a.ish + (long long)ix * a.isw + c), sc, sh,
                         a.tin.relu);
          lh[((bz * BY + by) * LP + bx) * VS + cc] = __builtin_bit_cast(unsigned short, (__bf16)v);
        }
      }
    } else {
    if (a.vec4) {
      constexpr int CV = KCI / 4;
      const int cv = tid % CV;           // 256 % CV == 0: fixed channel group per thread
      const int c = c0 + cv * 4;
      float sc[4], sh[4];
      nl_coeff_vec<4>(a.tin, n, a.Ci, c, sc, sh);
      constexpr int U = 4, STEP = 256 / CV;
      for (int bv0 = tid / CV; bv0 < boxvox; bv0 += U * STEP) {
        float4 xin[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int bv = bv0 + u * STEP;
          const int bz = (int)__umulhi((unsigned)bv, ci.mBXY), brem = bv - bz * ci.BXY;
          const int by = (int)__umulhi((unsigned)brem, ci.mBX), bx = brem - by * ci.BX;
          const int iz = iz0 + bz, iy = iy0 + by, ix = ix0 + bx;
          ok[u] = bv < boxvox && (unsigned)iz < (unsigned)a.Di && (unsigned)iy < (unsigned)a.Hi &&
                  (unsigned)ix < (unsigned)a.Wi && c < a.Ci;
          xin[u] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (ok[u]) xin[u] = *reinterpret_cast<const float4*>(inb + inoff + iz * a.isd + iy * a.ish + ix * a.isw + c);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int bv = bv0 + u * STEP;
          if (bv < boxvox) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok[u]) {
              v.x = nl_apply(xin[u].x, sc[0], sh[0], a.tin.relu);
              v.y = (c + 1 < a.Ci) ? nl_apply(xin[u].y, sc[1], sh[1], a.tin.relu) : 0.f;
              v.z = (c + 2 < a.Ci) ? nl_apply(xin[u].z, sc[2], sh[2], a.tin.relu) : 0.f;
              v.w = (c + 3 < a.Ci) ? nl_apply(xin[u].w, sc[3], sh[3], a.tin.relu) : 0.f;
            }
            float* d = lds + bv * VS + cv * 4;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
          }
        }
      }
    } else {
      const int cc = tid % KCI;          // 256 % KCI == 0
      const int c = c0 + cc;
      float sc, sh;
      nl_coeff_vec<1>(a.tin, n, a.Ci, c, &sc, &sh);
      for (int bv = tid / KCI; bv < boxvox; bv += 256 / KCI) {
        const int bx = bv % BX, by = (bv / BX) % BY, bz = bv / (BX * BY);
        const int iz = iz0 + bz, iy = iy0 + by, ix = ix0 + bx;
        float v = 0.f;
        if ((unsigned)iz < (unsigned)a.Di && (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi &&
            c < a.Ci)
          v = nl_apply(inb[inoff + iz * a.isd + iy * a.ish + ix * a.isw + c], sc, sh, a.tin.relu);
        lds[bv * VS + cc] = v;
      }
    }
    }
    __syncthreads();

    // ---------------- MFMA over taps x channel blocks ----------------
    if constexpr (BF) {
      if (colact) {
        constexpr int KS = KCI / 16;
        const unsigned short* lh = reinterpret_cast<const unsigned short*>(lds);
        const int kreal = min(KCI, a.Ci - c0);
        const int nks = (kreal + 15) >> 4;         // 16-channel steps that carry data
        const uint4* wq = reinterpret_cast<const uint4*>(wpn);   // image [tap][Kp/8][Np][8 bf16]
        const long long slabsz8 = (long long)(a.Kp / 8) * a.Np;
        const uint4* wcol = wq + (long long)(c0 / 8 + h) * a.Np + colbase + r;
        const int np2 = 2 * a.Np;
        const int ntap = ci.ntaps;
        const int* tslab = a.slab + ci.tap0;
        const int* ttoff = a.toff + ci.tap0;
        if (nks == KS && ntap == 27) {
          // Full 27-tap stage: the weight fragments of G taps are fetched as a group while the previous group's
          // G*KS*MB MFMAs run - two register sets, straight-line code, scheduling barriers so the loads stay ahead
          // (the scheduler otherwise sinks each load next to its use and every tap exposes an L2 round trip:
          // measured 15-18k cycles per stage against 3.5k cycles of MFMA).
          constexpr int G = OCC > 2 ? 3 : (KS == 1 ? 9 : 5);
          constexpr int NG = (27 + G - 1) / G;
          static_assert(G == WG0 && KS == WKS, "first-group prefetch must match the group shape");
          // canonical stage (row loader): tap offsets are immediates, weight requests scalar-based (tap27_off, wfrag27);
          // otherwise both come from the tap tables
          auto stage27 = [&](auto fc) {
          constexpr bool CANON = decltype(fc)::value;
          uint4 wset[2][G][KS];
#pragma unroll
          for (int t = 0; t < G; ++t)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) wset[0][t][ks] = wfirst[t][ks];     // requested before the staging pass
          uint4 avc[MB];
          {
            const int ta0 = CANON ? 0 : ttoff[0] * VS;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) avc[mb] = *reinterpret_cast<const uint4*>(lh + rowaddr[mb] + ta0);
          }
#pragma unroll
          for (int g = 0; g < NG; ++g) {
            if (g + 1 < NG) {
#pragma unroll
              for (int t = 0; t < G; ++t) {
                if ((g + 1) * G + t < 27) {
#pragma unroll
                  for (int ks = 0; ks < KS; ++ks)
                    wset[(g + 1) & 1][t][ks] = CANON ? wfrag27(c0, (g + 1) * G + t, ks) : wcol[tslab[(g + 1) * G + t] * slabsz8 + ks * np2];
                }
              }
            }
            if (g == NG - 2 && pipe && stage + 1 < ks1) issue(stage + 1);     // behind the stage's last weight request
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < G; ++t) {
              if (g * G + t < 27) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                  const bf16x8 bfr = __builtin_bit_cast(bf16x8, wset[g & 1][t][ks]);
                  // the activation fragments of the NEXT (tap, k-step) are read while this one's MFMAs run: left to
                  // itself the compiler emits read -> wait -> MFMA per fragment (one register set) and every MFMA
                  // exposes an LDS round trip (measured ~93 cycles per 32-cycle MFMA)
                  const int idx = g * G + t;
                  const bool more = !(idx == 26 && ks == KS - 1);
                  const int nidx = ks + 1 < KS ? idx : idx + 1, nks2 = ks + 1 < KS ? ks + 1 : 0;
                  uint4 avn[MB];
                  int tan = 0;
                  if (more) tan = (CANON ? tap27_off(nidx, RBY) : ttoff[nidx]) * VS + nks2 * 16;
#pragma unroll
                  for (int mb = 0; mb < MB; ++mb) {
                    if (more) avn[mb] = *reinterpret_cast<const uint4*>(lh + rowaddr[mb] + tan);
                    acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, avc[mb]), bfr, acc[mb], 0, 0, 0);
                  }
#pragma unroll
                  for (int mb = 0; mb < MB; ++mb) {
                    if (more) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);    // one LDS read ...
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);              // ... then one MFMA
                  }
                  if (more) {
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb) avc[mb] = avn[mb];
                  }
                }
              }
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          };
          if constexpr (CANON_CFG) {
            if (fast) stage27(std::true_type{});
            else stage27(std::false_type{});
          } else {
            stage27(std::false_type{});
          }
        } else if (nks == KS && ntap >= 4) {
          // A tap is only MB*KS MFMAs of 32 cycles: far less than an L2 round trip, so the weight fragments run
          // through a ring of D taps in flight.  The body is branch-free (a load behind a branch is waited for on
          // the spot): the tap count is padded to a multiple of D, a padded tap multiplies by a zero fragment and
          // re-reads a valid LDS / weight address.
          constexpr int D = 4;
          uint4 ring[D][KS];
#pragma unroll
          for (int d = 0; d < D; ++d) {
            const uint4* wb = wcol + tslab[d] * slabsz8;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) ring[d][ks] = wb[ks * np2];
          }
          const int ntap_pad = (ntap + D - 1) / D * D;
          for (int tp0 = 0; tp0 < ntap_pad; tp0 += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
              const int tp = tp0 + d;
              const bool live = tp < ntap;
              bf16x8 bfrag[KS];
#pragma unroll
              for (int ks = 0; ks < KS; ++ks) {
                uint4 q = ring[d][ks];
                q.x = live ? q.x : 0u; q.y = live ? q.y : 0u; q.z = live ? q.z : 0u; q.w = live ? q.w : 0u;
                bfrag[ks] = __builtin_bit_cast(bf16x8, q);
              }
              const uint4* wb = wcol + tslab[min(tp + D, ntap - 1)] * slabsz8;
#pragma unroll
              for (int ks = 0; ks < KS; ++ks) ring[d][ks] = wb[ks * np2];
              const int ta = ttoff[min(tp, ntap - 1)] * VS;
#pragma unroll
              for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                  const uint4 av = *reinterpret_cast<const uint4*>(lh + rowaddr[mb] + ta + ks * 16);
                  acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), bfrag[ks], acc[mb], 0, 0, 0);
                }
              }
            }
          }
        } else {
          for (int tp = 0; tp < ntap; ++tp) {
            const uint4* wb = wcol + tslab[tp] * slabsz8;
            const int ta = ttoff[tp] * VS;
            for (int ks = 0; ks < nks; ++ks) {
              const bf16x8 bfrag = __builtin_bit_cast(bf16x8, wb[ks * np2]);
#pragma unroll
              for (int mb = 0; mb < MB; ++mb) {
                const uint4 av = *reinterpret_cast<const uint4*>(lh + rowaddr[mb] + ta + ks * 16);
                acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), bfrag, acc[mb], 0, 0, 0);
              }
            }
          }
        }
      }
    } else {
    // ---------------- MFMA over taps x channel pairs ----------------
    if (colact) {
      constexpr int KK = KCI / 2;
      const int kreal = min(KCI, a.Ci - c0);
      const float* wcol = wpn + (long long)c0 * a.Np + colbase + r + (long long)h * a.Np;
      const long long slabsz = (long long)a.Kp * a.Np;
      const int np2 = 2 * a.Np;
      const int ntap = ci.ntaps;
      const int* tslab = a.slab + ci.tap0;
      const int* ttoff = a.toff + ci.tap0;
      if (kreal == KCI) {
        // full stage: branch-free body, weight fragments one tap ahead
        float bcur[KK], bnxt[KK];
        {
          const float* wb = wcol + tslab[0] * slabsz;
#pragma unroll
          for (int kk = 0; kk < KK; ++kk) bcur[kk] = wb[kk * np2];
        }
        for (int tp = 0; tp < ntap; ++tp) {
          const int tn = min(tp + 1, ntap - 1);
          const float* wb = wcol + tslab[tn] * slabsz;
#pragma unroll
          for (int kk = 0; kk < KK; ++kk) bnxt[kk] = wb[kk * np2];
          const int ta = ttoff[tp] * VS;
#pragma unroll
          for (int kk = 0; kk < KK; ++kk) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
              const float av = lds[rowaddr[mb] + ta + 2 * kk];
              acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bcur[kk], acc[mb], 0, 0, 0);
            }
          }
#pragma unroll
          for (int kk = 0; kk < KK; ++kk) bcur[kk] = bnxt[kk];
        }
      } else {
        // tail stage (Cin not a multiple of KCI): only the channel pairs that carry data
        const int kkn = (kreal + 1) >> 1;
        for (int tp = 0; tp < ntap; ++tp) {
          const float* wb = wcol + tslab[tp] * slabsz;
          const int ta = ttoff[tp] * VS;
          for (int kk = 0; kk < kkn; ++kk) {
            const float b = wb[kk * np2];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
              const float av = lds[rowaddr[mb] + ta + 2 * kk];
              acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b, acc[mb], 0, 0, 0);
            }
          }
        }
      }
    }
    }
    __syncthreads();
  }

  // ---------------- epilogue ----------------
  const int col = colbase + r;
  const bool colok = colact && col < a.Co;
  float s_sum = 0.f, s_sq = 0.f;
  if (a.ksplit > 1) {
    if (colact) {
      float* wsb = a.ws + ((long long)lbz * gridDim.x + bx) * MT * a.Np + col;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
          const int v = (mg * MB + mb) * 32 + row;
          wsb[(long long)v * a.Np] = acc[mb][i];
        }
    }
    return;
  }
  if (a.ovec) {
    // 16-byte stores through a wave-private LDS tile (the K loop ended with a barrier: the box image is dead)
    float v_sum[4], v_sq[4];
    epilogue_vec16<TZ, TY, TX, MB, BF, ABF>(a, biasn, ci, acc, lds + wave * EPI_TILE_FLOATS, lane, mg * MB, colbase, colact, n, gz0,
                                            gy0, gx0, v_sum, v_sq);
    if (a.stats != nullptr) {
      float* red = lds + 4 * EPI_TILE_FLOATS;      // [4 waves][2][32], behind the four tiles
      if (lane < 8) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          red[(wave * 2 + 0) * 32 + lane * 4 + j] = v_sum[j];
          red[(wave * 2 + 1) * 32 + lane * 4 + j] = v_sq[j];
        }
      }
      __syncthreads();
      if (mg == 0 && h == 0 && colok) {
        float ts = 0.f, tq = 0.f;
#pragma unroll
        for (int g = 0; g < MG; ++g) {
          ts += red[((g * NB + cb) * 2 + 0) * 32 + r];
          tq += red[((g * NB + cb) * 2 + 1) * 32 + r];
        }
        const long long row = (long long)n * a.stats_rows_per_n + cidx * (a.tz * a.ty * a.tx) + tile_in_n;
        a.stats[(row * 2 + 0) * a.Co + col] = ts;
        a.stats[(row * 2 + 1) * a.Co + col] = tq;
      }
    }
    return;
  }
  float bias = 0.f, asc = 1.f, ash = 0.f;
  if (colok) {
    if (biasn) bias = biasn[col];
    if (a.add) nl_coeff(a.tadd, n, a.Co, col, asc, ash);
  }
  // Per 32-row block: addresses and masks of its 16 rows first, then ALL loads of the fused add / accumulate
  // (unconditional, from clamped addresses, under wave-uniform branches), then the arithmetic and the stores: a load
  // inside a per-element branch would be waited for on the spot, 16 exposed round trips per block.
  const int colc = min(col, a.Co - 1);
  const float* addb = a.add;
  float* outb = a.out;
  const long long abase = (long long)n * a.asn + colc, obase = (long long)n * a.osn + colc;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {       // 8 rows at a time: keeps the kernel at two waves per SIMD
      int ooff[8], aoff[8];                      // element offsets inside batch item n (host checks < 2^31)
      bool ok[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int i = half * 8 + j;
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        int zl, yl, xl;
        row_to_local<TZ, TY, TX, BF>((mg * MB + mb) * 32 + row, zl, yl, xl);
        const int gz = gz0 + zl, gy = gy0 + yl, gx = gx0 + xl;
        const int oz = gz * a.so + ci.oz, oy = gy * a.so + ci.oy, ox = gx * a.so + ci.ox;
        ok[j] = colok && gz < ci.Dg && gy < ci.Hg && gx < ci.Wg && oz < a.Do && oy < a.Ho && ox < a.Wo;
        const int cz = min(oz, a.Do - 1), cy = min(oy, a.Ho - 1), cx = min(ox, a.Wo - 1);
        ooff[j] = cz * (int)a.osd + cy * (int)a.osh + cx * (int)a.osw;
        aoff[j] = cz * (int)a.asd + cy * (int)a.ash + cx * (int)a.asw;
      }
      float addv[8], oldv[8];
      if (a.add) {
#pragma unroll
        for (int j = 0; j < 8; ++j) addv[j] = ld1_t<ABF>(addb, abase + aoff[j]);
      }
      if (a.accumulate) {
#pragma unroll
        for (int j = 0; j < 8; ++j) oldv[j] = ld1_t<ABF>(outb, obase + ooff[j]);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float v = acc[mb][half * 8 + j] + bias;
        if (a.add) v += nl_apply(addv[j], asc, ash, a.tadd.relu);
        if (a.accumulate) v += oldv[j];
        if (ok[j]) {
          st1_t<ABF>(outb, obase + ooff[j], v);
          s_sum += v;
          s_sq += v * v;
        }
      }
    }
  }
  if (a.stats != nullptr) {
    // lanes l and l+32 hold the same column; waves of different m-groups too.
    s_sum += __shfl_xor(s_sum, 32, 64);
    s_sq += __shfl_xor(s_sq, 32, 64);
    float* red = lds;  // [4 waves][2][32]; safe: the K loop ended with a barrier
    if (h == 0) {
      red[(wave * 2 + 0) * 32 + r] = s_sum;
      red[(wave * 2 + 1) * 32 + r] = s_sq;
    }
    __syncthreads();
    if (mg == 0 && h == 0 && colok) {
      float ts = 0.f, tq = 0.f;
#pragma unroll
      for (int g = 0; g < MG; ++g) {
        ts += red[((g * NB + cb) * 2 + 0) * 32 + r];
        tq += red[((g * NB + cb) * 2 + 1) * 32 + r];
      }
      const long long row = (long long)n * a.stats_rows_per_n + cidx * (a.tz * a.ty * a.tx) + tile_in_n;
      a.stats[(row * 2 + 0) * a.Co + col] = ts;
      a.stats[(row * 2 + 1) * a.Co + col] = tq;
    }
  }
}

// ---------------------------------------------------------------- class-fused stride-2 transposed forms (bf16 operands)
// ConvTranspose3d forward and the input gradient of a stride-2 Conv3d are 8 output parity classes, each a dense
// convolution over 1 / 2 / 4 / 8 taps of the coarse input grid (27 taps in all).  igemm_kernel runs them as 8 x tiles
// independent workgroups that each stage their own halo box for 1-8 taps: per 16-channel stage 4-32 MFMAs per wave
// against a full staging pass and two barriers (measured: convT 128->32 at 64^3 out = 70 us at 105 TFLOP/s, the slowest
// implicit-GEMM launches of the network).  Here ONE workgroup produces all 8 classes of a coarse 4 x 4 x 8 tile (= an
// 8 x 8 x 16 block of the output): the box (tile + 1 on the high side of every axis: all tap offsets are 0 or 1) is staged
// once, the 27 taps' weight fragments of the stage go through LDS too (every wave needs all of them), and wave w owns
// row block w (z slice w of the tile) of EVERY class - 27 MFMAs per k step and wave, perfectly balanced, fed by 8
// activation fragments (one per offset, shared by the classes) and 27 weight fragments from LDS.
struct ClsTap { int cls, d, slab; };
__host__ __device__ constexpr ClsTap cls_tap(int f) {        // f-th (class, tap) in class order; d = (dz*2 + dy)*2 + dx
  int idx = 0;
  for (int cls = 0; cls < 8; ++cls) {
    const int pz = (cls >> 2) & 1, py = (cls >> 1) & 1, px = cls & 1;
    for (int a = 0; a <= pz; ++a)
      for (int b = 0; b <= py; ++b)
        for (int c = 0; c <= px; ++c) {
          // parity 0: k = 1 reads g (d = 0); parity 1: k = 0 reads g + 1, k = 2 reads g   (build_taps)
          const int kz = pz ? (a == 0 ? 0 : 2) : 1, dz = pz ? (a == 0 ? 1 : 0) : 0;
          const int ky = py ? (b == 0 ? 0 : 2) : 1, dy = py ? (b == 0 ? 1 : 0) : 0;
          const int kx = px ? (c == 0 ? 0 : 2) : 1, dx = px ? (c == 0 ? 1 : 0) : 0;
          if (idx == f) return ClsTap{cls, (dz * 2 + dy) * 2 + dx, (kz * 3 + ky) * 3 + kx};
          ++idx;
        }
  }
  return ClsTap{0, 0, 0};
}

template <int KCI, bool ABF>
__global__ __launch_bounds__(256, 2) void igemm_cls8_kernel(GArgs a) {
  constexpr int TZ = 4, TY = 4, TX = 8, VS = KCI + 8, LP = LDS_PITCH_BF16, KS = KCI / 16, KC8 = KCI / 8;
  constexpr int BZ = TZ + 1, BY = TY + 1, BX = TX + 1;
  constexpr int BOX_HW = BZ * BY * LP * VS;                 // halfwords of the box image
  constexpr int WIMG = 27 * KC8 * 32;                       // uint4 entries of the stage's weight image
  extern __shared__ float lds[];
  unsigned short* lh = reinterpret_cast<unsigned short*>(lds);
  uint4* wimg = reinterpret_cast<uint4*>(lh + (BOX_HW + 7) / 8 * 8);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, r = lane & 31;
  int t = blockIdx.x;
  const int tile_in_n = t % (a.tz * a.ty * a.tx);
  const int txi = t % a.tx; t /= a.tx;
  const int tyi = t % a.ty; t /= a.ty;
  const int tzi = t % a.tz;
  const int n = t / a.tz;
  const float* biasn = pset_bias(a.ps, a.bias, n);
  const int gz0 = tzi * TZ, gy0 = tyi * TY, gx0 = txi * TX;
  const int colbase = blockIdx.y * 32;
  const bool colact = colbase < a.Np;

  int zl, yl, xl;
  row_to_local<TZ, TY, TX, true>(wave * 32 + r, zl, yl, xl);
  const unsigned short* arow = lh + ((zl * BY + yl) * LP + xl) * VS + 8 * h;
  const uint4* brow = wimg + h * 32 + r;

  f32x16 acc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;

  const uint4* wq = reinterpret_cast<const uint4*>(pset_packed(a.ps, a.wp, n));
  const long long slabsz8 = (long long)(a.Kp / 8) * a.Np;
  const unsigned isd = (unsigned)a.isd, ish = (unsigned)a.ish, isw = (unsigned)a.isw;
  const int cmax = ABF ? ((a.Ci - 1) & ~7) : ((a.Ci - 1) & ~3);
  constexpr int AIT = (BZ * BY * BX * KC8 + 255) / 256;      // box items per thread
  constexpr int BIT = (WIMG + 255) / 256;                    // weight entries per thread
  const float relu_lo = a.tin.relu ? 0.f : -__builtin_inff();

  for (int ks = 0; ks < a.nstages; ++ks) {
    const int c0 = ks * KCI;
    const float* sb = reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.in) + ((long long)n * a.isn + c0) * (ABF ? 2 : 4));
    // ---- requests first: weights of the stage, norm-on-load coefficients, box items (one round trip for all of them)
    uint4 wv[BIT];
#pragma unroll
    for (int j = 0; j < BIT; ++j) {
      const int i = min(tid + 256 * j, WIMG - 1);
      const int tap = i / (KC8 * 32), rem = i - tap * (KC8 * 32), k8l = rem >> 5, col = rem & 31;
      wv[j] = wq[tap * slabsz8 + (long long)(c0 / 8 + k8l) * a.Np + colbase + col];
    }
    Oct8<ABF> av[AIT];
    unsigned aok = 0u;
    int alds[AIT];
#pragma unroll
    for (int j = 0; j < AIT; ++j) {
      const int i = min(tid + 256 * j, BZ * BY * BX * KC8 - 1);
      const int cv = i % KC8, bv = i / KC8;
      const int bz = bv / (BY * BX), brem = bv - bz * (BY * BX), by = brem / BX, bx = brem - by * BX;
      const int iz = gz0 + bz, iy = gy0 + by, ix = gx0 + bx;
      const bool ok = iz < a.Di && iy < a.Hi && ix < a.Wi;
      aok |= (ok ? 1u : 0u) << j;
      const unsigned off = __umul24((unsigned)min(iz, a.Di - 1), isd) + __umul24((unsigned)min(iy, a.Hi - 1), ish) +
                           __umul24((unsigned)min(ix, a.Wi - 1), isw) + cv * 8;
      const int over = max(c0 + cv * 8 - cmax, 0), over_hi = max(c0 + cv * 8 + 4 - cmax, 0);
      av[j] = oct8_ld<ABF>(sb, off - over, off + 4 - over_hi);
      alds[j] = ((bz * BY + by) * LP + bx) * VS + cv * 8;
    }
    float sc[AIT][8], sh[AIT][8];
#pragma unroll
    for (int j = 0; j < AIT; ++j) {
      const int i = min(tid + 256 * j, BZ * BY * BX * KC8 - 1);
      nl_coeff_vec<8>(a.tin, n, a.Ci, c0 + (i % KC8) * 8, sc[j], sh[j]);
    }
    // ---- commit
#pragma unroll
    for (int j = 0; j < BIT; ++j)
      if (tid + 256 * j < WIMG) wimg[tid + 256 * j] = wv[j];
#pragma unroll
    for (int j = 0; j < AIT; ++j) {
      float v[8];
      oct8_f8(av[j], v);
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = fmaxf(fmaf(v[q], sc[j][q], sh[j][q]), relu_lo);
      const unsigned okm = ((aok >> j) & 1u) ? 0xffffffffu : 0u;
      uint4 pk;
      pk.x = pack_bf16x2(v[0], v[1]) & okm; pk.y = pack_bf16x2(v[2], v[3]) & okm;
      pk.z = pack_bf16x2(v[4], v[5]) & okm; pk.w = pack_bf16x2(v[6], v[7]) & okm;
      if (tid + 256 * j < BZ * BY * BX * KC8) *reinterpret_cast<uint4*>(lh + alds[j]) = pk;
    }
    __syncthreads();
    // ---- MFMAs: 8 activation fragments per k step (one per tap offset), then the 27 (class, tap) products; weight
    // fragment f is read LA products ahead through a ring (a fence per MFMA keeps the reads where they are written)
    if (colact) {
#pragma unroll
      for (int k16 = 0; k16 < KS; ++k16) {
        uint4 af[8];
#pragma unroll
        for (int d = 0; d < 8; ++d)
          af[d] = *reinterpret_cast<const uint4*>(arow + ((((d >> 2) & 1) * BY + ((d >> 1) & 1)) * LP + (d & 1)) * VS + k16 * 16);
        constexpr int LA = 3;
        uint4 bring[LA + 1];
        static_for<0, LA>([&](auto fc) {
          constexpr int f = decltype(fc)::value;
          bring[f] = brow[(cls_tap(f).slab * KC8 + k16 * 2) * 32];
        });
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, 27>([&](auto fc) {
          constexpr int f = decltype(fc)::value;
          constexpr ClsTap ct = cls_tap(f);
          if constexpr (f + LA < 27) bring[(f + LA) % (LA + 1)] = brow[(cls_tap(f + LA).slab * KC8 + k16 * 2) * 32];
          acc[ct.cls] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[ct.d]),
                                                                __builtin_bit_cast(bf16x8, bring[f % (LA + 1)]), acc[ct.cls], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        });
      }
    }
    __syncthreads();
  }
  // ---- epilogue: every class's 32 x 32 block through this wave's LDS transposition tile (the box image is dead)
  float tsum[4] = {0.f, 0.f, 0.f, 0.f}, tsq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    float v_sum[4], v_sq[4];
    epilogue_vec16<TZ, TY, TX, 1, true, ABF>(a, biasn, a.cls[c], *reinterpret_cast<f32x16(*)[1]>(&acc[c]), lds + wave * EPI_TILE_FLOATS,
                                             lane, wave, colbase, colact, n, gz0, gy0, gx0, v_sum, v_sq);
#pragma unroll
    for (int j = 0; j < 4; ++j) { tsum[j] += v_sum[j]; tsq[j] += v_sq[j]; }
  }
  if (a.stats != nullptr) {
    float* red = lds + 4 * EPI_TILE_FLOATS;      // [4 waves][2][32], behind the four tiles
    if (lane < 8) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        red[(wave * 2 + 0) * 32 + lane * 4 + j] = tsum[j];
        red[(wave * 2 + 1) * 32 + lane * 4 + j] = tsq[j];
      }
    }
    __syncthreads();
    const int col = colbase + tid;
    if (tid < 32 && col < a.Co) {
      float ts = 0.f, tq = 0.f;
#pragma unroll
      for (int g = 0; g < 4; ++g) { ts += red[(g * 2 + 0) * 32 + tid]; tq += red[(g * 2 + 1) * 32 + tid]; }
      const long long row = (long long)n * a.stats_rows_per_n + tile_in_n;
      a.stats[(row * 2 + 0) * a.Co + col] = ts;
      a.stats[(row * 2 + 1) * a.Co + col] = tq;
    }
  }
}

template <int KCI, bool ABF>
static int launch_cls8_t(const GArgs& a, int tiles, hipStream_t s) {
  constexpr int VS = KCI + 8;
  size_t lds = ((size_t)(5 * 5 * LDS_PITCH_BF16 * VS + 7) / 8 * 8) * 2 + (size_t)27 * (KCI / 8) * 32 * 16;
  const size_t epi = (4 * EPI_TILE_FLOATS + 4 * 2 * 32) * sizeof(float);
  if (lds < epi) lds = epi;
  auto kern = igemm_cls8_kernel<KCI, ABF>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(tiles, a.Np / 32), dim3(256), lds, s, a);
  return launch_status("conv igemm (class-fused stride-2 form)");
}

// Reduce split-K slabs: sum over splits, then the same epilogue as above.
// grid (tiles * MT/32, ceil(Np/32)); 256 threads = 8 row lanes x 32 columns, 4 rows per thread, i.e. one
// block per 32 rows x 32 columns; each block writes ONE statistics row (rows per tile = MT/32).
template <int TZ, int TY, int TX, bool PERM, bool ABF = false>
__global__ __launch_bounds__(256) void splitk_finalize_kernel(GArgs a, int tiles) {
  __shared__ float red[2][8][32];
  constexpr int MT = TZ * TY * TX;
  constexpr int RB = MT / 32;
  const int tid = threadIdx.x, r = tid & 31, rg = tid >> 5;
  const int tile = blockIdx.x / RB, rb = blockIdx.x % RB;
  int cidx, t;
  class_of_tile(a, tile, cidx, t);
  const ClassInfo ci = a.cls[cidx];
  const int tile_in_n = t % (a.tz * a.ty * a.tx);
  const int txi = t % a.tx; t /= a.tx;
  const int tyi = t % a.ty; t /= a.ty;
  const int tzi = t % a.tz;
  const int n = t / a.tz;
  const int col = blockIdx.y * 32 + r;
  const bool colok = col < a.Co;
  float bias = 0.f, asc = 1.f, ash = 0.f;
  if (colok) {
    const float* biasn = pset_bias(a.ps, a.bias, n);
    if (biasn) bias = biasn[col];
    if (a.add) nl_coeff(a.tadd, n, a.Co, col, asc, ash);
  }
  float s_sum = 0.f, s_sq = 0.f;
  // the thread's 4 rows advance together: every load is unconditional (clamped address) and independent
  const int colc = min(col, a.Co - 1);
  const long long kstride = (long long)tiles * MT * a.Np;
  const float* wsp[4];
  long long ooff[4], aoff[4];
  bool ok[4];
  float s4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int v = rb * 32 + rg * 4 + q;
    int zl, yl, xl;
    row_to_local<TZ, TY, TX, PERM>(v, zl, yl, xl);
    const int gz = tzi * TZ + zl, gy = tyi * TY + yl, gx = txi * TX + xl;
    const int oz = gz * a.so + ci.oz, oy = gy * a.so + ci.oy, ox = gx * a.so + ci.ox;
    ok[q] = colok && gz < ci.Dg && gy < ci.Hg && gx < ci.Wg && oz < a.Do && oy < a.Ho && ox < a.Wo;
    const int cz = min(oz, a.Do - 1), cy = min(oy, a.Ho - 1), cx = min(ox, a.Wo - 1);
    ooff[q] = (long long)n * a.osn + cz * a.osd + cy * a.osh + cx * a.osw + colc;
    aoff[q] = (long long)n * a.asn + cz * a.asd + cy * a.ash + cx * a.asw + colc;
    wsp[q] = a.ws + ((long long)tile * MT + v) * a.Np + min(col, a.Np - 1);
  }
  for (int kz = 0; kz < a.ksplit; kz += 4) {        // 4 splits x 4 rows = 16 independent loads per trip
    float v[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q][u] = wsp[q][(long long)min(kz + u, a.ksplit - 1) * kstride];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int q = 0; q < 4; ++q) s4[q] += kz + u < a.ksplit ? v[q][u] : 0.f;
  }
  float addv[4] = {0.f, 0.f, 0.f, 0.f}, oldv[4] = {0.f, 0.f, 0.f, 0.f};
  if (a.add) {
#pragma unroll
    for (int q = 0; q < 4; ++q) addv[q] = ld1_t<ABF>(a.add, aoff[q]);
  }
  if (a.accumulate) {
#pragma unroll
    for (int q = 0; q < 4; ++q) oldv[q] = ld1_t<ABF>(a.out, ooff[q]);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float val = s4[q] + bias;
    if (a.add) val += nl_apply(addv[q], asc, ash, a.tadd.relu);
    if (a.accumulate) val += oldv[q];
    if (ok[q]) {
      st1_t<ABF>(a.out, ooff[q], val);
      s_sum += val;
      s_sq += val * val;
    }
  }
  if (a.stats != nullptr) {
    red[0][rg][r] = s_sum;
    red[1][rg][r] = s_sq;
    __syncthreads();
    if (rg == 0 && colok) {
      float ts = 0.f, tq = 0.f;
#pragma unroll
      for (int g = 0; g < 8; ++g) { ts += red[0][g][r]; tq += red[1][g][r]; }
      const long long row = (long long)n * a.stats_rows_per_n + (cidx * (a.tz * a.ty * a.tx) + tile_in_n) * RB + rb;
      a.stats[(row * 2 + 0) * a.Co + col] = ts;
      a.stats[(row * 2 + 1) * a.Co + col] = tq;
    }
  }
}

// ---------------------------------------------------------------- weight packing
// torch layout W[A][B][T] (A = dim0, B = dim1, T = k^3 taps) -> P[T][Kp][Np]
// kn_is_ba: K = B, N = A (Conv3d forward, ConvTranspose3d input gradient)
// else    : K = A, N = B (Conv3d input gradient, ConvTranspose3d forward)
// W'[d][k][(p, co)] of the 2x2x2 gather-GEMM up-convolution (conv_direct.hip): tap (kz, ky, kx) of a k3 s2 transposed
// convolution belongs to exactly one (coarse offset d, output parity p) pair - per axis k = 1: (p 0, d 0); k = 0: (p 1, d 1);
// k = 2: (p 1, d 0) - the image is in MFMA B-fragment order (lane = 32 * ((k >> 3) & 1) + column, 8 k per lane), bf16; the
// 37 slots no tap owns are written as zeros.  One call writes the 64 (d, p) entries of one (input channel k, output channel
// co) pair; `src` = that pair's 27 taps (null: a padded output channel, all zeros).
__device__ __forceinline__ void upconv8_put_pair(unsigned short* img, int K, int k, int co, const float* src) {
  for (int dp = 0; dp < 64; ++dp) {
    const int d = dp >> 3, p = dp & 7;
    int tap = 0;
    bool owned = true;
#pragma unroll
    for (int ax = 2; ax >= 0; --ax) {                       // z, y, x: (p 0, d 0) -> k 1; (p 1, d 1) -> k 0; (p 1, d 0) -> k 2
      const int pa = (p >> ax) & 1, da = (d >> ax) & 1;
      owned = owned && !(pa == 0 && da == 1);
      tap = tap * 3 + (pa == 0 ? 1 : (da == 1 ? 0 : 2));
    }
    const float v = (owned && src != nullptr) ? src[tap] : 0.f;
    img[(((d * (K >> 4) + (k >> 4)) * 64 + ((k >> 3) & 1) * 32 + p * 4 + co) << 3) + (k & 7)] =
        (unsigned short)(f32x2_to_bf16x2(v, 0.f) & 0xffffu);
  }
}

// the whole W' image of one ConvTranspose3d weight [K][N][27] (single-layer pack path)
__global__ void pack_upconv8_kernel(const float* __restrict__ w, unsigned short* __restrict__ img, int K, int N) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= K * 4) return;
  const int k = i >> 2, co = i & 3;
  upconv8_put_pair(img, K, k, co, co < N ? w + ((long long)k * N + co) * 27 : nullptr);
}

__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ p, int A, int B, int T,
                                    int Kp, int Np, int kn_is_ba) {
  const long long total = (long long)T * Kp * Np;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int nn = (int)(i % Np);
    const int k = (int)((i / Np) % Kp);
    const int tp = (int)(i / ((long long)Np * Kp));
    const int K = kn_is_ba ? B : A, N = kn_is_ba ? A : B;
    float v = 0.f;
    if (k < K && nn < N) {
      const int ai = kn_is_ba ? nn : k, bi = kn_is_ba ? k : nn;
      v = w[((long long)ai * B + bi) * T + tp];
    }
    p[i] = v;
  }
}

// bf16 image [T][Kp/8][Np][8]: a lane's MFMA B fragment (8 consecutive k for one column) is one 16-byte load
__global__ void pack_weights_bf16_kernel(const float* __restrict__ w, unsigned short* __restrict__ p, int A, int B, int T,
                                         int Kp, int Np, int kn_is_ba) {
  const long long total = (long long)T * Kp * Np;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(i & 7);
    const int nn = (int)((i >> 3) % Np);
    const int k8 = (int)((i / (8LL * Np)) % (Kp / 8));
    const int tp = (int)(i / ((long long)Kp * Np));
    const int k = k8 * 8 + j;
    const int K = kn_is_ba ? B : A, N = kn_is_ba ? A : B;
    float v = 0.f;
    if (k < K && nn < N) {
      const int ai = kn_is_ba ? nn : k, bi = kn_is_ba ? k : nn;
      v = w[((long long)ai * B + bi) * T + tp];
    }
    p[i] = __builtin_bit_cast(unsigned short, (__bf16)v);
  }
}

// All packed images of a model in ONE launch: a device table of entries; a workgroup finds its entry by
// binary search over the prefix sums of the entries' workgroup counts (uniform, scalar).
//
// A packed image is a transpose of the torch weight (taps innermost there, outermost here), so a workgroup
// moves one 8 (k) x 32 (n) x T tile through LDS: coalesced runs of the master in, 128-byte (fp32: one n row)
// or 512-byte (bf16: 32 n x 8 k) runs of the image out.  ~155 MB per U-Net repack: HBM-bound.
struct PackEntry {
  const float* w; void* p;
  unsigned short* up8;      // fragment-ordered bf16 image of the gather-GEMM up-convolution (or null)
  uint4* chfr; int KI, NO;  // B-fragment image of the thin-K matrix-core convolution (or null), its actual K and N
  int A, B, T, Kp, Np, kn_is_ba, bf16, direct;
  long long start;   // first workgroup of this entry
};

constexpr int PK = 8, PN = 32;

template <int T>
__device__ __forceinline__ void pack_tile(const PackEntry& e, int lb, float* tile) {
  const int tiles_n = e.Np / PN;
  const int k0 = (lb / tiles_n) * PK, n0 = (lb % tiles_n) * PN;
  const int K = e.kn_is_ba ? e.B : e.A, N = e.kn_is_ba ? e.A : e.B;
  constexpr int RS = PK * T + 1;                 // row stride of the [n][k*T+tap] staging (odd: no bank conflicts)
  // PN*PK*T is a multiple of 256 (6912 = 27 * 256, 256): U loads are issued before the first one is stored (one load per
  // trip left the workgroup waiting 27 memory latencies in a row)
  constexpr int IT = PN * PK * T / 256, U = IT % 9 == 0 ? 9 : 1;
  static_assert(PN * PK * T % 256 == 0, "pack tile: whole trips");
  // full 27-tap tiles whose runs start on 16 bytes (B a multiple of 4: every wide layer): the runs are read as float4,
  // 7 loads per thread instead of 27 and a quarter of the index arithmetic (the kernel was vector-ALU bound on it)
  constexpr int RUN4 = (T == 27) ? PK * T / 4 : 1;            // float4 per n-run (kn_is_ba) = 54
  const bool full = T == 27 && k0 + PK <= K && n0 + PN <= N && e.B % 4 == 0 && ((uintptr_t)e.w) % 16 == 0;
  if (T == 27 && full) {
    constexpr int N4 = PN * PK * T / 4, IT4 = (N4 + 255) / 256;      // 1728 float4, 7 trips (the last one partial)
    float4 v[IT4];
    if (e.kn_is_ba) {
#pragma unroll
      for (int u = 0; u < IT4; ++u) {
        const int i4 = min(u * 256 + (int)threadIdx.x, N4 - 1);
        const int nn = i4 / RUN4, r4 = i4 % RUN4;
        v[u] = *reinterpret_cast<const float4*>(e.w + ((long long)(n0 + nn) * e.B + k0) * T + r4 * 4);
      }
#pragma unroll
      for (int u = 0; u < IT4; ++u) {
        const int i4 = u * 256 + (int)threadIdx.x;
        if (i4 < N4) {
          const int nn = i4 / RUN4, r4 = i4 % RUN4;
          float* d = tile + nn * RS + r4 * 4;                 // RS is odd: four 4-byte stores
          d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w;
        }
      }
    } else {
      constexpr int KRUN4 = PN * T / 4;                       // float4 per k-run = 216
#pragma unroll
      for (int u = 0; u < IT4; ++u) {
        const int i4 = min(u * 256 + (int)threadIdx.x, N4 - 1);
        const int kk = i4 / KRUN4, r4 = i4 % KRUN4;
        v[u] = *reinterpret_cast<const float4*>(e.w + ((long long)(k0 + kk) * e.B + n0) * T + r4 * 4);
      }
#pragma unroll
      for (int u = 0; u < IT4; ++u) {
        const int i4 = u * 256 + (int)threadIdx.x;
        if (i4 < N4) *reinterpret_cast<float4*>(tile + i4 * 4) = v[u];
      }
    }
  } else
  if (e.kn_is_ba) {                              // a = n, b = k: per n a run of PK*T contiguous floats
    for (int it = 0; it < IT; it += U) {
      float v[U]; int dst[U]; bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int idx = (it + u) * 256 + threadIdx.x;
        const int nn = idx / (PK * T), r = idx % (PK * T);
        const int kk = r / T;
        // unconditional load from a clamped address, masked afterwards (a load behind a branch is waited for at once)
        ok[u] = n0 + nn < N && k0 + kk < K;
        v[u] = e.w[((long long)min(n0 + nn, N - 1) * e.B + min(k0 + kk, K - 1)) * T + (r - kk * T)];
        dst[u] = nn * RS + r;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) tile[dst[u]] = ok[u] ? v[u] : 0.f;
    }
  } else {                                       // a = k, b = n: per k a run of PN*T contiguous floats
    for (int it = 0; it < IT; it += U) {
      float v[U]; bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int idx = (it + u) * 256 + threadIdx.x;
        const int kk = idx / (PN * T), r = idx % (PN * T);
        const int nn = r / T;
        ok[u] = k0 + kk < K && n0 + nn < N;
        v[u] = e.w[((long long)min(k0 + kk, K - 1) * e.B + min(n0 + nn, N - 1)) * T + (r - nn * T)];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) tile[(it + u) * 256 + threadIdx.x] = ok[u] ? v[u] : 0.f;
    }
  }
  __syncthreads();
  const long long slab = (long long)e.Kp * e.Np;
  if (e.bf16) {
    for (int o = threadIdx.x; o < T * PN; o += 256) {
      const int nn = o % PN, tp = o / PN;
      unsigned short h[8];
#pragma unroll
      for (int kk = 0; kk < PK; ++kk) {
        const float v = e.kn_is_ba ? tile[nn * RS + kk * T + tp] : tile[kk * (PN * T) + nn * T + tp];
        h[kk] = __builtin_bit_cast(unsigned short, (__bf16)v);
      }
      uint4 q;
      q.x = h[0] | ((unsigned)h[1] << 16); q.y = h[2] | ((unsigned)h[3] << 16);
      q.z = h[4] | ((unsigned)h[5] << 16); q.w = h[6] | ((unsigned)h[7] << 16);
      *reinterpret_cast<uint4*>((unsigned short*)e.p + tp * slab + ((long long)(k0 >> 3) * e.Np + n0 + nn) * 8) = q;
    }
  } else {
    for (int o = threadIdx.x; o < T * PK * PN; o += 256) {
      const int nn = o % PN, kk = (o / PN) % PK, tp = o / (PN * PK);
      const float v = e.kn_is_ba ? tile[nn * RS + kk * T + tp] : tile[kk * (PN * T) + nn * T + tp];
      ((float*)e.p)[tp * slab + (long long)(k0 + kk) * e.Np + n0 + nn] = v;
    }
  }
}

// B fragments of the thin-K matrix-core convolution from the layer's fp32 tap image [27][Kp][Np] (conv_direct.hip:
// chan_frag_bytes has the layout).  Runs behind the kernel that wrote the tap image.
__device__ __forceinline__ void chan_frags_write(const float* img, uint4* fr, int Kp, int Np, int KI, int N) {
  const int NB = N / 32;
  for (int i = threadIdx.x; i < 7 * NB * 64; i += blockDim.x) {
    const int lane = i & 63, frn = i >> 6, s2 = frn / NB, nb = frn % NB;
    const int h = lane >> 5, col = nb * 32 + (lane & 31);
    float wv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int tap = s2 * 4 + h * 2 + (e >> 2), c = e & 3;
      wv[e] = (tap < 27 && c < KI && col < N) ? img[((long long)tap * Kp + c) * Np + col] : 0.f;
    }
    uint4 pk;
    pk.x = f32x2_to_bf16x2(wv[0], wv[1]); pk.y = f32x2_to_bf16x2(wv[2], wv[3]);
    pk.z = f32x2_to_bf16x2(wv[4], wv[5]); pk.w = f32x2_to_bf16x2(wv[6], wv[7]);
    fr[i] = pk;
  }
}
__global__ __launch_bounds__(256) void pack_chan_frags_kernel(const float* img, uint4* fr, int Kp, int Np, int KI, int N) {
  chan_frags_write(img, fr, Kp, Np, KI, N);
}
__global__ __launch_bounds__(256) void pack_chan_frags_batched_kernel(const PackEntry* __restrict__ tab) {
  const PackEntry e = tab[blockIdx.x];
  if (e.chfr != nullptr) chan_frags_write((const float*)e.p, e.chfr, e.Kp, e.Np, e.KI, e.NO);
}

__global__ __launch_bounds__(256) void pack_batched_kernel(const PackEntry* __restrict__ tab, int count) {
  __shared__ __attribute__((aligned(16))) float tile[PN * (PK * 27 + 1)];
  int lo = 0, hi = count - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].start <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const PackEntry e = tab[lo];
  const int lb = (int)(blockIdx.x - e.start);
  if (e.direct) {
    // [T][K][4] fp32 image of a direct-convolution layer: a few KB, one position per thread
    const int li = lb * 256 + threadIdx.x;
    if (li >= e.Kp * e.Np) return;
    const int nn = li % e.Np, k = li / e.Np;
    const int N = e.kn_is_ba ? e.A : e.B;
    const int ai = e.kn_is_ba ? nn : k, bi = e.kn_is_ba ? k : nn;
    const float* src = e.w + ((long long)ai * e.B + bi) * e.T;
    for (int tp = 0; tp < e.T; ++tp)
      ((float*)e.p)[(long long)tp * e.Kp * e.Np + li] = nn < N ? src[tp] : 0.f;
    if (e.up8 != nullptr) upconv8_put_pair(e.up8, e.Kp, k, nn, nn < N ? src : nullptr);      // + the gather-GEMM image W'
    return;
  }
  if (e.T == 27) pack_tile<27>(e, lb, tile);
  else pack_tile<1>(e, lb, tile);
}


// ------------------------------------------------------------------ streaming 1x1x1 convolution (bf16 operands, round 3)
// The 1x1x1 layers of the deep-fusion decoder at full resolution (33 -> 32 shortcut convolutions and their input gradients at
// 128^3, 96 -> 64 at 64^3) ran on the gather GEMM above: LDS box, barriers and a 27-tap tile shape for what is a [voxels x K] x
// [K x N] product - 3.3 to 9 times their HBM time.  Without taps there is nothing to share between voxels, so nothing goes
// through LDS: lane (r, h) of a wave loads the 8 channels k16 * 16 + 8 h .. + 7 of voxel r STRAIGHT into its A fragment (one
// 16-byte load of bf16, two of fp32), the B fragments of all k steps sit in registers for the whole launch (K <= 128, N <=
// 64), and a wave owns 32 voxels per trip.  Epilogue as in chan_mfma_kernel: bias, fused add (with its norm-on-load),
// accumulate, bf16 rows stored as channel-pair dwords.  No statistics, no norm-on-load of the input: those calls stay above.
struct PWArgs {
  const float* in; long long isn; unsigned isw; int Ci;       // voxel-dense tensors: element offset of voxel v = v * sw
  float* out; long long osn; unsigned osw; int Co;
  const float* add; long long asn; unsigned asw; NL tadd;
  const float* wp; const float* bias; int Np; PSets ps;
  int accumulate; long long dhw;
};

template <int KB, int NB, bool BF>
__global__ __launch_bounds__(256) void pointwise_mfma_kernel(PWArgs a) {
  constexpr int TP = NB * 32 + 4;                     // tile row pitch (floats): 16-byte rows, conflict-free column writes
  __shared__ __attribute__((aligned(16))) float tile[4][32][TP];
  __shared__ float coef[3][NB * 32];                  // bias | scale, shift of the fused add, per output column
  const int n = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, r = lane & 31;
  const float* wpn = pset_packed(a.ps, a.wp, n);
  const float* biasn = pset_bias(a.ps, a.bias, n);
  uint4 bfr[KB][NB];
  {
    const uint4* wq = reinterpret_cast<const uint4*>(wpn);            // [Kp / 8][Np] entries of 8 bf16
#pragma unroll
    for (int ks = 0; ks < KB; ++ks)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) bfr[ks][nb] = wq[(long long)(ks * 2 + h) * a.Np + nb * 32 + r];
  }
  if (tid < NB * 32) {
    const int col = min(tid, a.Co - 1);
    float sc = 1.f, sh = 0.f;
    if (a.add) nl_coeff(a.tadd, n, a.Co, col, sc, sh);
    coef[0][tid] = (biasn && tid < a.Co) ? biasn[col] : 0.f;
    coef[1][tid] = sc; coef[2][tid] = sh;
  }
  __syncthreads();
  const float* inn = item_base<BF>(a.in, n, a.isn);
  float* outn = const_cast<float*>(item_base<BF>(a.out, n, a.osn));
  const float* addn = a.add == nullptr ? nullptr : item_base<BF>(a.add, n, a.asn);
  const float relu_lo = a.tadd.relu ? 0.f : -__builtin_inff();
  // every wave makes the same number of trips (a wave past the end works on clamped voxels and stores nothing): the tile
  // hand-over below needs no more than the wave's own program order
  for (long long base = (long long)blockIdx.x * 128; base < a.dhw; base += (long long)gridDim.x * 128) {
    const long long v0 = base + wave * 32;
    const unsigned vo = (unsigned)min(v0 + r, a.dhw - 1) * a.isw;
    uint4 af[KB];
#pragma unroll
    for (int ks = 0; ks < KB; ++ks) {
      const unsigned c0 = ks * 16 + h * 8;
      if (c0 + 8 <= a.isw) {          // (rows are padded to 8 channels; what lies behind the row is not read)
        if constexpr (BF) {
          af[ks] = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(inn) + vo + c0);
        } else {
          const float4 lo = *reinterpret_cast<const float4*>(inn + vo + c0), hi = *reinterpret_cast<const float4*>(inn + vo + c0 + 4);
          af[ks] = make_uint4(f32x2_to_bf16x2(lo.x, lo.y), f32x2_to_bf16x2(lo.z, lo.w), f32x2_to_bf16x2(hi.x, hi.y), f32x2_to_bf16x2(hi.z, hi.w));
        }
      } else if (!BF && c0 + 4 <= a.isw) {          // fp32 rows are padded to 4 channels
        const float4 lo = *reinterpret_cast<const float4*>(inn + vo + c0);
        af[ks] = make_uint4(f32x2_to_bf16x2(lo.x, lo.y), f32x2_to_bf16x2(lo.z, lo.w), 0u, 0u);
      } else {
        af[ks] = make_uint4(0u, 0u, 0u, 0u);
      }
      if ((int)c0 + 8 > a.Ci) {          // channels past the tensor's last one (a slice of a wider row, pad lanes): exact zeros
        auto pm = [&](int c) { return (c < a.Ci ? 0xffffu : 0u) | (c + 1 < a.Ci ? 0xffff0000u : 0u); };
        af[ks].x &= pm(c0); af[ks].y &= pm(c0 + 2); af[ks].z &= pm(c0 + 4); af[ks].w &= pm(c0 + 6);
      }
    }
    f32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KB; ++ks)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[ks]), __builtin_bit_cast(bf16x8, bfr[ks][nb]), acc[nb], 0, 0, 0);
    // ---- epilogue through the wave's LDS tile: accumulator i of lane (h, r) = voxel (i & 3) + 8 (i >> 2) + 4 h, column
    // nb * 32 + r goes in column-wise (conflict-free), comes back as 8 consecutive channels of one voxel per lane: the fused
    // add and the accumulate operand are one 16-byte load each, the result one 16-byte store
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) tile[wave][(i & 3) + 8 * (i >> 2) + 4 * h][nb * 32 + r] = acc[nb][i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int q = 0; q < 2 * NB; ++q) {
      const int item = lane + 64 * q, vl = item / (4 * NB), c8 = item % (4 * NB), c0 = c8 * 8;
      const long long vv = v0 + vl;
      if (vv < a.dhw && c0 < a.Co) {
        const float4 t0 = *reinterpret_cast<const float4*>(&tile[wave][vl][c0]), t1 = *reinterpret_cast<const float4*>(&tile[wave][vl][c0 + 4]);
        float v[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
        const bool full = c0 + 8 <= a.Co;
        const unsigned oo = (unsigned)vv * a.osw + c0;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += coef[0][c0 + e];
        if (a.add) {
          float ad[8];
          if (full || c0 + 8 <= (int)a.asw) {
            oct8_f8(oct8_ld<BF>(addn, (unsigned)vv * a.asw + c0, (unsigned)vv * a.asw + c0 + 4), ad);
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) ad[e] = c0 + e < a.Co ? ld1_t<BF>(addn, (unsigned)vv * a.asw + c0 + e) : 0.f;
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += fmaxf(fmaf(ad[e], coef[1][c0 + e], coef[2][c0 + e]), relu_lo);
        }
        if (full) {
          if (a.accumulate) {
            float old[8];
            oct8_f8(oct8_ld<BF>(outn, oo, oo + 4), old);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += old[e];
          }
          oct8_st<BF>(outn, oo, v);
        } else {                           // ragged last chunk: element stores, nothing behind the last channel is touched
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (c0 + e < a.Co) st1_t<BF>(outn, oo + e, a.accumulate ? v[e] + ld1_t<BF>(outn, oo + e) : v[e]);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

template <int KB, bool BF>
static void launch_pointwise_mfma_nb(const PWArgs& a, int nb, dim3 grid, hipStream_t s) {
  if (nb == 1) hipLaunchKernelGGL((pointwise_mfma_kernel<KB, 1, BF>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((pointwise_mfma_kernel<KB, 2, BF>), grid, dim3(256), 0, s, a);
}
template <bool BF>
static void launch_pointwise_mfma(const PWArgs& a, int kb, int nb, dim3 grid, hipStream_t s) {
  switch (kb) {
    case 1: launch_pointwise_mfma_nb<1, BF>(a, nb, grid, s); break;
    case 2: launch_pointwise_mfma_nb<2, BF>(a, nb, grid, s); break;
    case 3: launch_pointwise_mfma_nb<3, BF>(a, nb, grid, s); break;
    case 4: launch_pointwise_mfma_nb<4, BF>(a, nb, grid, s); break;
    case 6: launch_pointwise_mfma_nb<6, BF>(a, nb, grid, s); break;
    default: launch_pointwise_mfma_nb<8, BF>(a, nb, grid, s); break;
  }
}

// ---------------------------------------------------------------- host side
struct Config { int NB, MB, TZ, TY, TX, KCI; bool bf; };

static inline int roundup(int v, int m) { return (v + m - 1) / m * m; }

static void op_dims(const mmtta_conv_desc* d, int& K, int& N, int& si, bool& classes) {
  switch (d->op) {
    case MMTTA_CONV_FWD: K = d->cin; N = d->cout; si = d->stride; classes = false; break;
    case MMTTA_CONV_DGRAD: K = d->cout; N = d->cin; si = 1; classes = d->stride == 2; break;
    case MMTTA_CONVT_FWD: K = d->cin; N = d->cout; si = 1; classes = true; break;
    default: K = d->cout; N = d->cin; si = 2; classes = false; break;  // CONVT_DGRAD
  }
}

static Config pick_config(int Np, int si, long long voxels, int K, bool bf) {
  // fp32 stage depth / bf16 stage depth (bf16 stages are multiples of the MFMA K = 16)
  if (si == 1) {
    if (Np == 32 && bf && g_igemm_lean) return {1, 2, 4, 8, 8, 16, bf};      // lean tile, four workgroups per CU
    if (Np == 32) return {1, 4, 8, 8, 8, bf ? 16 : 8, bf};
    // (128-voxel tiles for the 64-channel stride-1 layers as well - {2,2,4,4,8,16} - measured 64.6 against 65.2 volumes/s)
    if (Np == 64) return {2, 4, 4, 8, 8, bf ? 32 : 16, bf};
    // wide layers on a small grid (the 8^3 / 16^3 levels): shallow stages so that split-K can reach
    // >= 256 workgroups; otherwise 32-channel stages (fewer barriers, fewer weight fetches)
    const long long tiles = (voxels + 127) / 128, colgroups = (Np + 127) / 128;
    if (tiles * colgroups * ((K + 31) / 32) < 384) return {4, 4, 4, 4, 8, bf ? 16 : 8, bf};
    return {4, 4, 4, 4, 8, 32, bf};
  }
  if (Np == 32) return {1, 1, 4, 4, 8, bf ? 16 : 8, bf};
  // (the stride-2 64-column layers as two 32-column groups - twice the workgroups, the halo staged twice - measured 65.0
  // against 65.5 volumes/s)
  // (round 3: the 32 -> 64 layer at 64^3 fetches 3.6x its algorithmic bytes by the PMC count - two 16-channel stages each
  // touch half of every 64-byte voxel row of a 9 x 9 x 17 box that the L2 does not hold in between.  One 32-channel stage
  // on the same tile (110 KB of LDS: one workgroup per CU) 200 us against 124 per group of 8; on a 2 x 4 x 8 tile 146: the
  // launch is bound by its latency chain, not by the fabric)
  if (Np == 64) return {2, 2, 4, 4, 8, bf ? 16 : 8, bf};
  return {4, 4, 4, 4, 8, bf ? 16 : 8, bf};
}

// bf16 operands only where the reduction is deep enough for the K=16 MFMA and the layer is not on the
// direct (<= 4 produced channels) path; everything else computes in fp32 whatever desc.dtype says.
static bool use_bf16(const mmtta_conv_desc* d, int K) {
  return d->dtype == MMTTA_BF16 && K >= 16 && !direct_applicable(d);
}

static int validate_desc(const mmtta_conv_desc* d) {
  MMTTA_CHECK(d != nullptr, MMTTA_ERR_INVALID, "conv: null desc");
  MMTTA_CHECK(d->op >= 0 && d->op <= 3, MMTTA_ERR_INVALID, "conv: bad op %d", d->op);
  MMTTA_CHECK(d->ksize == 1 || d->ksize == 3, MMTTA_ERR_UNSUPPORTED, "conv: ksize %d (1 or 3)", d->ksize);
  MMTTA_CHECK(d->stride == 1 || d->stride == 2, MMTTA_ERR_UNSUPPORTED, "conv: stride %d (1 or 2)", d->stride);
  MMTTA_CHECK(!(d->ksize == 1 && d->stride != 1), MMTTA_ERR_UNSUPPORTED, "conv: 1x1x1 with stride 2");
  MMTTA_CHECK(d->cin > 0 && d->cout > 0, MMTTA_ERR_INVALID, "conv: channels must be positive");
  MMTTA_CHECK(d->dtype == MMTTA_F32 || d->dtype == MMTTA_BF16, MMTTA_ERR_UNSUPPORTED, "conv: dtype %d", d->dtype);
  if (d->op == MMTTA_CONVT_FWD || d->op == MMTTA_CONVT_DGRAD)
    MMTTA_CHECK(d->ksize == 3 && d->stride == 2, MMTTA_ERR_UNSUPPORTED,
                "conv_transpose: only k3 s2 p1 op1 (the monai UNet up layer)");
  return MMTTA_OK;
}

struct Geometry {
  int K, N, Kp, Np, si;
  bool classes;
  Config cfg;
  int tz, ty, tx, tiles_per_n, tiles, ncls;
  int nstages, ksplit, sps;
  int launches;
  bool fused;     // class-fused kernel of the stride-2 transposed forms (its own tiling: coarse 4 x 4 x 8, one row per tile)
};

static int expected_out_dim(const mmtta_conv_desc* d, int in) {
  switch (d->op) {
    case MMTTA_CONV_FWD: return d->stride == 1 ? in : (in + 1) / 2;
    case MMTTA_CONVT_FWD: return in * 2;
    case MMTTA_CONVT_DGRAD: return in / 2;
    default: return -1;  // CONV_DGRAD: caller gives dx shape (in*1 or in*2 or in*2-1)
  }
}

static int geometry(const mmtta_conv_desc* d, const mmtta_tensor* x, const mmtta_tensor* y, Geometry& g) {
  int st = validate_desc(d);
  if (st) return st;
  MMTTA_CHECK(x && y && x->ptr && y->ptr, MMTTA_ERR_INVALID, "conv: null tensor");
  MMTTA_CHECK(is_cl(x) && is_cl(y), MMTTA_ERR_UNSUPPORTED, "conv: tensors must be channels-last (sc == 1)");
  op_dims(d, g.K, g.N, g.si, g.classes);
  MMTTA_CHECK(x->c == g.K && y->c == g.N, MMTTA_ERR_INVALID, "conv: channel mismatch (x.c=%d want %d, y.c=%d want %d)",
              x->c, g.K, y->c, g.N);
  MMTTA_CHECK(x->n == y->n, MMTTA_ERR_INVALID, "conv: batch mismatch");
  const int xd[3] = {x->d, x->h, x->w}, yd[3] = {y->d, y->h, y->w};
  for (int i = 0; i < 3; ++i) {
    int e = expected_out_dim(d, xd[i]);
    if (e >= 0) {
      MMTTA_CHECK(yd[i] == e, MMTTA_ERR_INVALID, "conv: spatial mismatch axis %d: in %d -> out %d, expected %d", i, xd[i],
                  yd[i], e);
      if (d->op == MMTTA_CONVT_DGRAD)
        MMTTA_CHECK(xd[i] % 2 == 0, MMTTA_ERR_INVALID, "conv_transpose dgrad: dy extent must be even");
    } else {
      // CONV_DGRAD: x = dy (conv output extent), y = dx (conv input extent)
      int want = d->stride == 1 ? yd[i] : (yd[i] + 1) / 2;
      MMTTA_CHECK(xd[i] == want, MMTTA_ERR_INVALID, "conv dgrad: dy extent %d does not match dx extent %d", xd[i], yd[i]);
    }
  }
  g.Kp = roundup(g.K, 32);
  g.Np = roundup(g.N, 32);
  int Dg = y->d, Hg = y->h, Wg = y->w;
  if (g.classes) { Dg = (y->d + 1) / 2; Hg = (y->h + 1) / 2; Wg = (y->w + 1) / 2; }
  // launch geometry is a function of ONE batch item's extent: every item is computed as if launched alone (the contract of
  // the per-item parameter sets, include/mmtta.h), the batch only multiplies the workgroups
  g.cfg = pick_config(g.Np, g.si, (long long)Dg * Hg * Wg * (g.classes ? 8 : 1), g.K, use_bf16(d, g.K));
  g.tz = (Dg + g.cfg.TZ - 1) / g.cfg.TZ;
  g.ty = (Hg + g.cfg.TY - 1) / g.cfg.TY;
  g.tx = (Wg + g.cfg.TX - 1) / g.cfg.TX;
  g.tiles_per_n = g.tz * g.ty * g.tx;
  g.ncls = g.classes ? 8 : 1;
  g.tiles = g.tiles_per_n * x->n * g.ncls;     // the parity classes share one launch
  g.nstages = (g.K + g.cfg.KCI - 1) / g.cfg.KCI;
  g.launches = 1;
  const int ncolgroups = (g.Np + 32 * g.cfg.NB - 1) / (32 * g.cfg.NB);
  const int wgs = g.tiles_per_n * g.ncls * ncolgroups;      // per batch item
  g.ksplit = 1;
  g.sps = g.nstages;
  // Split the reduction when a launch has fewer than 192 workgroups, up to ~256.  Measured on the U-Net: with ONE volume
  // in flight 384/512 is the optimum (1024: +6 %, 256: +2 %, no split-K: +80 % of a step's conv time); with TWO in flight
  // (method.lanes: 2, the default) the other lane fills idle CUs and less splitting wins: 384/512, 192/256, 96/128 ->
  // 41.5, 42.9, 42.2 volumes/s (one lane at 192/256: 28.9 vs 29.6).
  if (wgs < g_tune[0] && g.nstages > 1) {
    int want = (g_tune[1] + wgs - 1) / wgs;
    if (want > g.nstages) want = g.nstages;
    g.sps = (g.nstages + want - 1) / want;
    g.ksplit = (g.nstages + g.sps - 1) / g.sps;
  }
  // Class-fused kernel (MMTTA_OPT_CLASS_FUSED_MIN_WORKGROUPS): all 8 parity classes of a coarse 4 x 4 x 8 tile in one
  // workgroup, when that still makes enough workgroups (no split-K there) and the tensors admit its 16-byte accesses
  g.fused = false;
  if (g.classes && g.cfg.bf && g_cls_fused_min > 0) {
    const int ftz = (Dg + 3) / 4, fty = (Hg + 3) / 4, ftx = (Wg + 7) / 8;
    const long long fw = (long long)ftz * fty * ftx * (g.Np / 32);      // per batch item
    auto al = [](const mmtta_tensor* t) {
      const int64_t q = is_bf16(t) ? 8 : 4, lim24 = (int64_t)1 << 24;
      const int64_t last = (int64_t)(t->d - 1) * t->sd + (int64_t)(t->h - 1) * t->sh + (int64_t)(t->w - 1) * t->sw + t->c + 16;
      return ((uintptr_t)t->ptr) % 16 == 0 && t->sw % q == 0 && t->sh % q == 0 && t->sd % q == 0 && t->sn % q == 0 &&
             t->sw < lim24 && t->sh < lim24 && t->sd < lim24 && last < ((int64_t)1 << 31);
    };
    if (fw >= g_cls_fused_min && al(x) && al(y) && y->c % 4 == 0 && is_bf16(x) == is_bf16(y) && g_epilogue_vec) {
      g.fused = true;
      g.tz = ftz; g.ty = fty; g.tx = ftx;
      g.tiles_per_n = ftz * fty * ftx;
      g.ncls = 1;                                  // one statistics row per fused tile
      g.tiles = g.tiles_per_n * x->n;
      g.nstages = (g.K + 15) / 16;
      g.ksplit = 1; g.sps = g.nstages;
    }
  }
  return MMTTA_OK;
}

// statistics rows written per tile: 1 by the conv epilogue, MT/32 by the split-K finalize
static int stats_rows_per_tile(const Geometry& g) {
  if (g.fused) return 1;
  return g.ksplit > 1 ? g.cfg.TZ * g.cfg.TY * g.cfg.TX / 32 : 1;
}

static void build_taps(const mmtta_conv_desc* d, int pz, int py, int px, Taps& t) {
  // pz/py/px: output parity class for the transposed forms, ignored otherwise
  int n = 0;
  const bool classes = (d->op == MMTTA_CONVT_FWD) || (d->op == MMTTA_CONV_DGRAD && d->stride == 2);
  if (d->ksize == 1) {
    t.dz[0] = t.dy[0] = t.dx[0] = 0; t.slab[0] = 0; n = 1;
  } else if (!classes) {
    // stride-1 input gradient mirrors the taps; they are enumerated in ascending (z, y, x) order of the voxel they READ in
    // both forms (kernel tap 26 - m for the mirrored one), the order the canonical stage of igemm_kernel assumes (tap27_off)
    const int sgn = (d->op == MMTTA_CONV_DGRAD) ? -1 : 1;
    for (int m = 0; m < 27; ++m) {
      const int q = sgn < 0 ? 26 - m : m, kz = q / 9, ky = q / 3 % 3, kx = q % 3;
      t.dz[n] = (signed char)(sgn * (kz - 1)); t.dy[n] = (signed char)(sgn * (ky - 1));
      t.dx[n] = (signed char)(sgn * (kx - 1)); t.slab[n] = q; ++n;
    }
  } else {
    // out index i = 2*o - 1 + k.  parity 0: k=1 reads o=g (d=0); parity 1: k=0 reads g+1, k=2 reads g.
    int kzs[2], dzs[2], nz, kys[2], dys[2], ny, kxs[2], dxs[2], nx;
    auto axis = [](int p, int* ks, int* ds) { if (p == 0) { ks[0] = 1; ds[0] = 0; return 1; }
                                              ks[0] = 0; ds[0] = 1; ks[1] = 2; ds[1] = 0; return 2; };
    nz = axis(pz, kzs, dzs); ny = axis(py, kys, dys); nx = axis(px, kxs, dxs);
    for (int a = 0; a < nz; ++a) for (int b = 0; b < ny; ++b) for (int c = 0; c < nx; ++c) {
      t.dz[n] = (signed char)dzs[a]; t.dy[n] = (signed char)dys[b]; t.dx[n] = (signed char)dxs[c];
      t.slab[n] = (kzs[a] * 3 + kys[b]) * 3 + kxs[c]; ++n;
    }
  }
  t.n = n;
  int zmn = 9, zmx = -9, ymn = 9, ymx = -9, xmn = 9, xmx = -9;
  for (int i = 0; i < n; ++i) {
    zmn = t.dz[i] < zmn ? t.dz[i] : zmn; zmx = t.dz[i] > zmx ? t.dz[i] : zmx;
    ymn = t.dy[i] < ymn ? t.dy[i] : ymn; ymx = t.dy[i] > ymx ? t.dy[i] : ymx;
    xmn = t.dx[i] < xmn ? t.dx[i] : xmn; xmx = t.dx[i] > xmx ? t.dx[i] : xmx;
  }
  t.zmin = zmn; t.ymin = ymn; t.xmin = xmn;
  t.zext = zmx - zmn; t.yext = ymx - ymn; t.xext = xmx - xmn;
}

template <int NB, int MB, int TZ, int TY, int TX, int KCI, bool BF, int OCC, bool ABF>
static int launch_cfg_t(const GArgs& a_in, const Taps* ht, int tiles, hipStream_t s) {
  GArgs a = a_in;
  size_t lds = 0;
  for (int c = 0; c < a.ncls; ++c) {
    const Taps& tp = ht[c];
    const int BZ = (TZ - 1) * a.si + tp.zext + 1, BY = (TY - 1) * a.si + tp.yext + 1, BX = (TX - 1) * a.si + tp.xext + 1;
    const int LP = (BF && a.si == 1) ? LDS_PITCH_BF16 : BX;
    for (int t = 0; t < tp.n; ++t) {
      a.toff[a.cls[c].tap0 + t] = ((tp.dz[t] - tp.zmin) * BY + (tp.dy[t] - tp.ymin)) * LP + (tp.dx[t] - tp.xmin);
      a.slab[a.cls[c].tap0 + t] = tp.slab[t];
    }
    a.cls[c].BX = BX; a.cls[c].BXY = BX * BY;
    a.cls[c].mBX = (unsigned)(((1ULL << 32) + BX - 1) / BX);
    a.cls[c].mBXY = (unsigned)(((1ULL << 32) + (unsigned long long)BX * BY - 1) / ((unsigned long long)BX * BY));
    const size_t need = BF ? (size_t)BZ * BY * LP * (KCI + 8) * 2 : (size_t)BZ * BY * BX * (KCI + 1) * sizeof(float);
    if (need > lds) lds = need;
  }
  if (BF && a.rowload && a.si == 1 && a.ncls == 1 && a.cls[0].ntaps == 27) {
    // the canonical stage addresses taps by formula: the tables must agree with it, or the generic loader runs
    a.flip27 = a.slab[0] == 26 ? 1 : 0;
    for (int t = 0; t < 27; ++t)
      if (a.toff[t] != tap27_off(t, TY + 2) || a.slab[t] != (a.flip27 ? 26 - t : t)) a.rowload = 0;
  }
  if (lds < (4 * EPI_TILE_FLOATS + 4 * 2 * 32) * sizeof(float)) lds = (4 * EPI_TILE_FLOATS + 4 * 2 * 32) * sizeof(float);
  lds = (lds + 15) / 16 * 16;
  a.coef_off = (int)(lds / sizeof(float));
  if (BF) lds += (size_t)2 * a.stages_per_split * KCI * sizeof(float);     // scale | shift of the staged channels
  MMTTA_CHECK(lds <= 160 * 1024, MMTTA_ERR_UNSUPPORTED, "conv: LDS box of %zu bytes exceeds 160 KiB", lds);
  int st;
  auto kern = igemm_kernel<NB, MB, TZ, TY, TX, KCI, BF, OCC, ABF>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  dim3 grid(tiles, (a.Np + 32 * NB - 1) / (32 * NB), a.ksplit);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
  st = launch_status("conv igemm");
  if (st) return st;
  if (a.ksplit > 1 && !g_profile_main_only) {
    dim3 g2(tiles * (TZ * TY * TX / 32), (a.Np + 31) / 32);
    hipLaunchKernelGGL((splitk_finalize_kernel<TZ, TY, TX, BF, ABF>), g2, dim3(256), 0, s, a, tiles);
    st = launch_status("conv split-K finalize");
  }
  return st;
}

// storage dispatch: bf16-stored activations exist for bf16-operand layers only (forward: input, output and fused add all
// bf16; everything else all fp32)
template <int NB, int MB, int TZ, int TY, int TX, int KCI, bool BF, int OCC = 2>
static int launch_cfg(const GArgs& a, const Taps* ht, int tiles, hipStream_t s) {
  if constexpr (BF) {
    if (a.in_bf || a.out_bf || a.add_bf) {
      MMTTA_CHECK(a.in_bf && a.out_bf && (a.add == nullptr || a.add_bf), MMTTA_ERR_UNSUPPORTED,
                  "conv: input, output and fused add must share one storage type (in %d out %d add %d)", a.in_bf, a.out_bf,
                  a.add_bf);
      return launch_cfg_t<NB, MB, TZ, TY, TX, KCI, BF, OCC, true>(a, ht, tiles, s);
    }
  }
  return launch_cfg_t<NB, MB, TZ, TY, TX, KCI, BF, OCC, false>(a, ht, tiles, s);
}

static int config_id(const Config& c) {
  int id;
  if (c.bf && c.NB == 1 && c.MB == 2) return 14;     // lean bf16 tile <1,2,4,8,8,16>
  if (c.NB == 1) id = c.MB == 4 ? 0 : 3;
  else if (c.NB == 2) id = c.MB == 4 ? 1 : 4;
  else id = c.KCI == 32 ? 2 : 5;
  return c.bf ? id + 7 : id;      // 6 = direct kernel; 7..12 = bf16 variants
}

static int launch_any(const Config& c, const GArgs& a, const Taps* ht, int tiles, hipStream_t s) {
  switch (config_id(c)) {
    case 0: return launch_cfg<1, 4, 8, 8, 8, 8, false>(a, ht, tiles, s);
    case 1: return launch_cfg<2, 4, 4, 8, 8, 16, false>(a, ht, tiles, s);
    case 2: return launch_cfg<4, 4, 4, 4, 8, 32, false>(a, ht, tiles, s);
    case 3: return launch_cfg<1, 1, 4, 4, 8, 8, false>(a, ht, tiles, s);
    case 4: return launch_cfg<2, 2, 4, 4, 8, 8, false>(a, ht, tiles, s);
    case 5: return launch_cfg<4, 4, 4, 4, 8, 8, false>(a, ht, tiles, s);
    case 14: return launch_cfg<1, 2, 4, 8, 8, 16, true, 3>(a, ht, tiles, s);
    case 7: return launch_cfg<1, 4, 8, 8, 8, 16, true>(a, ht, tiles, s);
    case 8: return launch_cfg<2, 4, 4, 8, 8, 32, true>(a, ht, tiles, s);
    case 9: return launch_cfg<4, 4, 4, 4, 8, 32, true>(a, ht, tiles, s);
    case 10: return launch_cfg<1, 1, 4, 4, 8, 16, true>(a, ht, tiles, s);
    case 11: return launch_cfg<2, 2, 4, 4, 8, 16, true>(a, ht, tiles, s);
    default: return launch_cfg<4, 4, 4, 4, 8, 16, true>(a, ht, tiles, s);
  }
}

}  // namespace mmtta

using namespace mmtta;

extern "C" int64_t mmtta_conv_packed_bytes(const mmtta_conv_desc* d) {
  if (validate_desc(d)) return -1;
  int K, N, si; bool cl;
  op_dims(d, K, N, si, cl);
  const int T = d->ksize * d->ksize * d->ksize;
  if (direct_applicable(d)) return (int64_t)T * K * 4 * (int64_t)sizeof(float) + upconv8_image_bytes(d);
  return (int64_t)T * roundup(K, 32) * roundup(N, 32) * (int64_t)(use_bf16(d, K) ? 2 : sizeof(float)) + chan_frag_bytes(d);
}

extern "C" int mmtta_conv_pack_weights(const mmtta_conv_desc* d, const float* w, void* packed, void* stream) {
  int st = validate_desc(d);
  if (st) return st;
  MMTTA_CHECK(w && packed, MMTTA_ERR_INVALID, "pack: null pointer");
  int K, N, si; bool cl;
  op_dims(d, K, N, si, cl);
  const int T = d->ksize * d->ksize * d->ksize;
  const bool convt = d->op == MMTTA_CONVT_FWD || d->op == MMTTA_CONVT_DGRAD;
  const int A = convt ? d->cin : d->cout, B = convt ? d->cout : d->cin;
  // K,N in terms of (A,B): CONV_FWD K=cin=B ; CONV_DGRAD K=cout=A ; CONVT_FWD K=cin=A ; CONVT_DGRAD K=cout=B
  const int kn_is_ba = (d->op == MMTTA_CONV_FWD || d->op == MMTTA_CONVT_DGRAD) ? 1 : 0;
  const bool direct = direct_applicable(d);
  const int Kp = direct ? K : roundup(K, 32), Np = direct ? 4 : roundup(N, 32);
  const long long total = (long long)T * Kp * Np;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  if (use_bf16(d, K))
    hipLaunchKernelGGL(pack_weights_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, (unsigned short*)packed,
                       A, B, T, Kp, Np, kn_is_ba);
  else
    hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, (float*)packed, A, B, T,
                       Kp, Np, kn_is_ba);
  if (direct && upconv8_image_bytes(d) > 0)       // ConvTranspose3d K -> R: + the gather-GEMM image behind the tap image
    hipLaunchKernelGGL(pack_upconv8_kernel, dim3((K * 4 + 255) / 256), dim3(256), 0, (hipStream_t)stream, w,
                       (unsigned short*)((char*)packed + (size_t)T * Kp * Np * 4), K, N);
  if (!direct && chan_frag_bytes(d) > 0)          // thin-K convolution: + its B fragments behind the fp32 tap image
    hipLaunchKernelGGL(pack_chan_frags_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)packed,
                       (uint4*)((char*)packed + (size_t)T * Kp * Np * 4), Kp, Np, K, N);
  return launch_status("pack weights");
}

static int fill_pack_entry(const mmtta_conv_desc* d, const float* w, void* packed, PackEntry& e) {
  int st = validate_desc(d);
  if (st) return st;
  MMTTA_CHECK(w && packed, MMTTA_ERR_INVALID, "pack: null pointer");
  int K, N, si; bool cl;
  op_dims(d, K, N, si, cl);
  const bool convt = d->op == MMTTA_CONVT_FWD || d->op == MMTTA_CONVT_DGRAD;
  const bool direct = direct_applicable(d);
  e.w = w; e.p = packed;
  e.A = convt ? d->cin : d->cout; e.B = convt ? d->cout : d->cin;
  e.T = d->ksize * d->ksize * d->ksize;
  e.Kp = direct ? K : roundup(K, 32); e.Np = direct ? 4 : roundup(N, 32);
  e.kn_is_ba = (d->op == MMTTA_CONV_FWD || d->op == MMTTA_CONVT_DGRAD) ? 1 : 0;
  e.bf16 = use_bf16(d, K) ? 1 : 0; e.direct = direct ? 1 : 0;
  e.up8 = (direct && upconv8_image_bytes(d) > 0) ? (unsigned short*)((char*)packed + (size_t)e.T * e.Kp * e.Np * 4) : nullptr;
  e.chfr = (!direct && chan_frag_bytes(d) > 0) ? (uint4*)((char*)packed + (size_t)e.T * e.Kp * e.Np * 4) : nullptr;
  e.KI = K; e.NO = N;
  MMTTA_CHECK(e.T == 1 || e.T == 27, MMTTA_ERR_UNSUPPORTED, "pack: ksize %d", d->ksize);
  e.start = 0;
  return MMTTA_OK;
}

extern "C" int64_t mmtta_conv_pack_table_bytes(int count) { return count < 0 ? -1 : (int64_t)count * (int64_t)sizeof(PackEntry); }

extern "C" int mmtta_conv_pack_table_build(const mmtta_pack_item* items, int count, void* table_host, int64_t* total) {
  MMTTA_CHECK(items && table_host && total && count > 0, MMTTA_ERR_INVALID, "pack table: bad argument");
  PackEntry* tab = (PackEntry*)table_host;
  long long run = 0;
  for (int i = 0; i < count; ++i) {
    int st = fill_pack_entry(&items[i].desc, items[i].w_master, items[i].packed, tab[i]);
    if (st) return st;
    tab[i].start = run;
    run += tab[i].direct ? ((long long)tab[i].Kp * tab[i].Np + 255) / 256
                         : (long long)(tab[i].Kp / PK) * (tab[i].Np / PN);
  }
  *total = run;
  return MMTTA_OK;
}

extern "C" int mmtta_conv_pack_batched(const void* table_dev, int count, int64_t total, void* stream) {
  MMTTA_CHECK(table_dev && count > 0 && total > 0, MMTTA_ERR_INVALID, "pack batched: bad argument");
  MMTTA_CHECK(total < (1LL << 31), MMTTA_ERR_UNSUPPORTED, "pack batched: %lld workgroups", (long long)total);
  hipLaunchKernelGGL(pack_batched_kernel, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, (const PackEntry*)table_dev,
                     count);
  int st = launch_status("pack batched");
  if (st) return st;
  // the thin-K layers' B fragments, from the tap images the launch above wrote (a block per table entry; most return at once)
  hipLaunchKernelGGL(pack_chan_frags_batched_kernel, dim3((unsigned)count), dim3(256), 0, (hipStream_t)stream, (const PackEntry*)table_dev);
  return launch_status("pack batched (thin-K fragments)");
}

extern "C" int mmtta_conv_plan(const mmtta_conv_desc* d, const mmtta_tensor* x, const mmtta_tensor* y,
                               mmtta_conv_plan_t* plan) {
  MMTTA_CHECK(plan != nullptr, MMTTA_ERR_INVALID, "conv plan: null plan");
  Geometry g;
  int st = geometry(d, x, y, g);
  if (st) return st;
  if (direct_applicable(d)) {
    plan->tiles = direct_blocks_per_n(d, x, y) * y->n;
    plan->launches = 1;
    plan->ksplit = 1;
    plan->stats_rows = plan->tiles;
    plan->config = 6;
    plan->_pad = 0;
    plan->workspace_bytes = 0;
    return MMTTA_OK;
  }
  if (chan_applicable(d, x, y) && !use_bf16(d, g.K)) {
    plan->tiles = chan_tiles_per_n(y) * y->n;
    plan->launches = 1;
    plan->ksplit = 1;
    plan->stats_rows = plan->tiles;
    plan->config = 13;
    plan->_pad = 0;
    plan->workspace_bytes = 0;
    return MMTTA_OK;
  }
  plan->tiles = g.tiles;
  plan->launches = g.launches;
  plan->ksplit = g.ksplit;
  plan->stats_rows = g.tiles * stats_rows_per_tile(g);
  plan->config = g.fused ? 15 : config_id(g.cfg);
  plan->_pad = 0;
  plan->workspace_bytes =
      g.ksplit > 1 ? (int64_t)g.ksplit * g.tiles * g.cfg.TZ * g.cfg.TY * g.cfg.TX * g.Np * (int64_t)sizeof(float) : 0;
  return MMTTA_OK;
}

extern "C" int mmtta_conv_run(const mmtta_conv_desc* d, const mmtta_tensor* x, const mmtta_norm_on_load* x_norm,
                              const void* packed, const float* bias, const mmtta_conv_epilogue* epi,
                              const mmtta_tensor* y, int accumulate, float* stats, void* workspace,
                              int64_t workspace_bytes, void* stream) {
  return mmtta_conv_run_sets(d, x, x_norm, packed, bias, epi, y, accumulate, stats, workspace, workspace_bytes, nullptr, stream);
}

extern "C" int mmtta_conv_run_sets(const mmtta_conv_desc* d, const mmtta_tensor* x, const mmtta_norm_on_load* x_norm,
                                   const void* packed, const float* bias, const mmtta_conv_epilogue* epi,
                                   const mmtta_tensor* y, int accumulate, float* stats, void* workspace,
                                   int64_t workspace_bytes, const mmtta_param_sets* sets, void* stream) {
  Geometry g;
  int st = geometry(d, x, y, g);
  if (st) return st;
  MMTTA_CHECK(packed != nullptr, MMTTA_ERR_INVALID, "conv: null packed weights");
  st = psets_validate(sets, x->n);
  if (st) return st;
  const PSets ps = psets(sets);
  if (direct_applicable(d)) return direct_conv_run(d, x, x_norm, packed, bias, epi, y, accumulate, stats, ps, (hipStream_t)stream);
  if (pointwise_small_applicable(d, x, y, stats, epi, x_norm) && !use_bf16(d, g.K) && is_f32(x))      // (y: fp32- or bf16-stored)
    return pointwise_small_run(x, packed, g.Kp, g.Np, bias, y, accumulate, ps, (hipStream_t)stream);
  if (chan_applicable(d, x, y) && !use_bf16(d, g.K))
    return chan_conv_run(d, x, x_norm, packed, g.Kp, g.Np, bias, epi, y, accumulate, stats, ps, (hipStream_t)stream);
  {  // 1x1x1 over many voxels with bf16 operands, nothing but bias / add / accumulate around it: the streaming kernel
    static const bool pw_on = !(getenv("MMTTA_POINTWISE_MFMA") && atoi(getenv("MMTTA_POINTWISE_MFMA")) == 0);      // (A/B switch)
    auto dense = [](const mmtta_tensor* t) { return t->sc == 1 && t->sh == (int64_t)t->w * t->sw && t->sd == (int64_t)t->h * t->sh; };
    const mmtta_tensor* ad = (epi && epi->add) ? epi->add : nullptr;
    const bool bfs = is_bf16(x);
    const long long dhw = (long long)y->d * y->h * y->w;
    const int kb = (g.K + 15) / 16;
    auto rows_ok = [&](const mmtta_tensor* t) {
      return dense(t) && is_bf16(t) == bfs && ((uintptr_t)t->ptr) % 16 == 0 && t->sw % (bfs ? 8 : 4) == 0 && t->sn % (bfs ? 8 : 4) == 0 &&
             dhw * t->sw < (1LL << 31);
    };
    const bool pw = pw_on && d->ksize == 1 && (d->op == MMTTA_CONV_FWD || d->op == MMTTA_CONV_DGRAD) && use_bf16(d, g.K) &&
                    (kb <= 4 || kb == 6 || kb == 8) && g.Np <= 64 && stats == nullptr && !(x_norm && (x_norm->mean || x_norm->scale)) &&
                    dhw >= 4096 && rows_ok(x) && rows_ok(y) && (ad == nullptr || (rows_ok(ad) && ad->n == y->n && ad->c == y->c && ad->d == y->d && ad->h == y->h && ad->w == y->w));
    if (pw) {
      PWArgs q;
      q.in = (const float*)x->ptr; q.isn = x->sn; q.isw = (unsigned)x->sw; q.Ci = x->c;
      q.out = (float*)y->ptr; q.osn = y->sn; q.osw = (unsigned)y->sw; q.Co = y->c;
      q.add = ad ? (const float*)ad->ptr : nullptr; q.asn = ad ? ad->sn : 0; q.asw = ad ? (unsigned)ad->sw : 0u;
      q.tadd = ad ? nl(&epi->add_norm) : nl(nullptr);
      q.wp = (const float*)packed; q.bias = bias; q.Np = g.Np; q.ps = ps; q.accumulate = accumulate; q.dhw = dhw;
      long long blocks = (dhw + 127) / 128;
      if (blocks > 2048) blocks = 2048;
      const dim3 grid((unsigned)blocks, y->n);
      if (bfs) launch_pointwise_mfma<true>(q, kb, g.Np / 32, grid, (hipStream_t)stream);
      else launch_pointwise_mfma<false>(q, kb, g.Np / 32, grid, (hipStream_t)stream);
      return launch_status("1x1 conv (streaming MFMA)");
    }
  }
  const int64_t need = g.ksplit > 1 ? (int64_t)g.ksplit * g.tiles * g.cfg.TZ * g.cfg.TY * g.cfg.TX * g.Np * 4 : 0;
  MMTTA_CHECK(need == 0 || (workspace != nullptr && workspace_bytes >= need), MMTTA_ERR_WORKSPACE,
              "conv: workspace %lld bytes, need %lld", (long long)workspace_bytes, (long long)need);
  GArgs a;
  a.in = (const float*)x->ptr; a.isn = x->sn; a.isd = x->sd; a.ish = x->sh; a.isw = x->sw;
  a.Ci = x->c; a.Di = x->d; a.Hi = x->h; a.Wi = x->w;
  a.tin = nl(x_norm);
  a.out = (float*)y->ptr; a.osn = y->sn; a.osd = y->sd; a.osh = y->sh; a.osw = y->sw;
  a.Co = y->c; a.Do = y->d; a.Ho = y->h; a.Wo = y->w;
  a.si = g.si;
  a.wp = (const float*)packed; a.Kp = g.Kp; a.Np = g.Np;
  a.bias = bias;
  a.ps = ps;
  a.add = nullptr; a.asn = a.asd = a.ash = a.asw = 0; a.tadd = nl(nullptr);
  if (epi && epi->add) {
    const mmtta_tensor* ad = epi->add;
    MMTTA_CHECK(ad->ptr && is_cl(ad) && ad->n == y->n && ad->c == y->c && ad->d == y->d && ad->h == y->h && ad->w == y->w,
                MMTTA_ERR_INVALID, "conv: epilogue `add` must be channels-last with the shape of y");
    a.add = (const float*)ad->ptr; a.asn = ad->sn; a.asd = ad->sd; a.ash = ad->sh; a.asw = ad->sw;
    a.tadd = nl(&epi->add_norm);
  }
  a.accumulate = accumulate;
  a.in_bf = is_bf16(x) ? 1 : 0;
  a.out_bf = is_bf16(y) ? 1 : 0;
  a.add_bf = (epi && epi->add && is_bf16(epi->add)) ? 1 : 0;
  {
    auto al16 = [](const void* p, long long sn, long long sd, long long sh, long long sw, int bf) {      // 4-channel accesses
      return ((uintptr_t)p) % (bf ? 8 : 16) == 0 && sn % 4 == 0 && sd % 4 == 0 && sh % 4 == 0 && sw % 4 == 0;
    };
    const bool oal = al16(y->ptr, y->sn, y->sd, y->sh, y->sw, a.out_bf) && y->c % 4 == 0 &&
                     (a.add == nullptr || al16(a.add, a.asn, a.asd, a.ash, a.asw, a.add_bf)) &&
                     (bias == nullptr || ((uintptr_t)bias) % 16 == 0);
    a.ovec = (oal && g_epilogue_vec) ? 1 : 0;
  }
  const int srt = stats_rows_per_tile(g);
  a.stats = stats; a.stats_rows_per_n = g.ncls * g.tiles_per_n * srt;
  a.ws = (float*)workspace; a.ksplit = g.ksplit; a.stages_per_split = g.sps; a.nstages = g.nstages;
  a.tz = g.tz; a.ty = g.ty; a.tx = g.tx;
  MMTTA_CHECK(g.cfg.bf || (!a.in_bf && !a.out_bf && !a.add_bf), MMTTA_ERR_UNSUPPORTED,
              "conv: bf16-stored tensors need a bf16-operand layer (K >= 16 in bf16 precision)");
  // 8-channel items: 32 bytes of fp32 (two 16-byte loads) or 16 bytes of bf16 (one): strides must keep them aligned
  const int am = a.in_bf ? 8 : 4;
  const bool al = (((uintptr_t)x->ptr) % 16 == 0) && x->sw % am == 0 && x->sh % am == 0 && x->sd % am == 0 && x->sn % am == 0;
  a.vec4 = al ? 1 : 0;
  a.flip27 = 0;
  {  // row-structured loader of the 3x3x3 stride-1 stages: 32-bit element offsets from 24-bit multiply-adds
    const int64_t lim24 = (int64_t)1 << 24;
    const int64_t last = (int64_t)(x->d - 1) * x->sd + (int64_t)(x->h - 1) * x->sh + (int64_t)(x->w - 1) * x->sw + x->c + 16;
    a.rowload = (al && x->sd < lim24 && x->sh < lim24 && x->sw < lim24 && x->d < lim24 && x->h < lim24 && x->w < lim24 &&
                 last < ((int64_t)1 << 31) && g_igemm_pipeline) ? 1 : 0;
  }
  Taps ht[8];
  a.ncls = g.classes ? 8 : 1;
  a.so = g.classes ? 2 : 1;
  int tap0 = 0;
  for (int cls = 0; cls < a.ncls; ++cls) {
    const int pz = (cls >> 2) & 1, py = (cls >> 1) & 1, px = cls & 1;
    build_taps(d, pz, py, px, ht[cls]);
    ClassInfo& ci = a.cls[cls];
    ci.tap0 = tap0; ci.ntaps = ht[cls].n; tap0 += ht[cls].n;
    ci.zmin = ht[cls].zmin; ci.ymin = ht[cls].ymin; ci.xmin = ht[cls].xmin;
    ci.zext = ht[cls].zext; ci.yext = ht[cls].yext; ci.xext = ht[cls].xext;
    if (g.classes) {
      ci.oz = pz; ci.oy = py; ci.ox = px;
      ci.Dg = (y->d - pz + 1) / 2; ci.Hg = (y->h - py + 1) / 2; ci.Wg = (y->w - px + 1) / 2;
    } else {
      ci.oz = ci.oy = ci.ox = 0;
      ci.Dg = y->d; ci.Hg = y->h; ci.Wg = y->w;
    }
  }
  if (g.fused) {
    MMTTA_CHECK(a.ovec && a.vec4, MMTTA_ERR_UNSUPPORTED,
                "conv (class-fused stride-2 form): the epilogue operands must admit 16-byte accesses");
    MMTTA_CHECK(a.in_bf == a.out_bf && (a.add == nullptr || a.add_bf == a.out_bf), MMTTA_ERR_UNSUPPORTED,
                "conv: input, output and fused add must share one storage type (in %d out %d add %d)", a.in_bf, a.out_bf, a.add_bf);
    return a.in_bf ? launch_cls8_t<16, true>(a, g.tiles, (hipStream_t)stream) : launch_cls8_t<16, false>(a, g.tiles, (hipStream_t)stream);
  }
  return launch_any(g.cfg, a, ht, g.tiles, (hipStream_t)stream);
}
