/*
 * mmtta.h - C ABI of libmmtta.so, the MI355X (gfx950) kernels of the per-volume adaptation
 * hot path of zhm1205/Multimodal_TTA.
 *
 * Boundary rules ("C-ABI layer"):
 *   - plain C: pointers, sizes, POD structs; no torch / C++ types in any signature;
 *   - every pointer is a DEVICE pointer owned by the caller; nothing is allocated or freed
 *     inside, all scratch memory is passed in as a workspace;
 *   - every call is asynchronous on the hipStream_t passed as `void* stream` (0 = default
 *     stream) and safe to capture into a hipGraph (no sync, no malloc, no host readback);
 *   - return value: 0 on success, negative mmtta_status otherwise; never throws.
 *
 * The reference has no native layer.  Each entry point names the torch / MONAI call it
 * stands in for and the reference file:line that reaches it.  "Reference" paths are relative
 * to the upstream repository root.
 *
 * Internal activation layout: channels-last NDHWC ("CL"), element stride 1 along C, described
 * by mmtta_tensor.  Boundary tensors (input volume, returned logits, labels) are NCDHW.
 */
#ifndef MMTTA_H
#define MMTTA_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMTTA_ABI_VERSION 2

typedef enum {
  MMTTA_OK = 0,
  MMTTA_ERR_INVALID = -1,     /* bad argument (null pointer, negative size, bad enum) */
  MMTTA_ERR_UNSUPPORTED = -2, /* valid request this build has no kernel for (stated in message) */
  MMTTA_ERR_LAUNCH = -3,      /* hipLaunch / hipGetLastError failure */
  MMTTA_ERR_WORKSPACE = -4    /* workspace too small */
} mmtta_status;

typedef enum { MMTTA_F32 = 0, MMTTA_BF16 = 1 } mmtta_dtype;

/* A 5-D view.  Strides are in ELEMENTS.  Kernels that need channels-last require sc == 1. */
typedef struct {
  void* ptr;
  int32_t n, c, d, h, w;
  int64_t sn, sc, sd, sh, sw;
  int32_t dtype; /* mmtta_dtype */
  int32_t flags; /* MMTTA_TENSOR_* */
} mmtta_tensor;

/* The (sw - c) elements behind every voxel's channels are padding that belongs to this view (a buffer allocated
 * with a channel row padded to 4 floats, not a slice of a wider tensor): kernels may overwrite them with zeros so
 * that a 3-channel voxel is ONE full 16-byte store instead of three partial ones (partial 32-byte sectors cost a
 * read-modify-write in HBM). */
#define MMTTA_TENSOR_OWNS_PAD 1

/* Per-(n,c) normalisation applied to a tensor WHEN IT IS READ ("norm on load"):
 *   v = (x - mean[n*C+c]) * rstd[n*C+c] * (gamma ? gamma[c] : 1) + (beta ? beta[c] : 0);
 *   if (relu) v = max(v, 0);
 * mean == NULL means "no transform".  This is MONAI's ADN (Norm -> Dropout(p=0) -> ReLU) of the
 * producing Convolution, folded into the consumer (reference:
 * src/models/unet_multimodal_midfusion.py:45-55 via monai Convolution/ADN). */
typedef struct {
  const float* mean;  /* [N*C] or NULL */
  const float* rstd;  /* [N*C] */
  const float* gamma; /* [C] or NULL */
  const float* beta;  /* [C] or NULL */
  int32_t relu;
  int32_t _pad;
  /* optional precombined form written by mmtta_norm_stats_finalize: v = x*scale[n*C+c] + shift[n*C+c].
   * When given, consumers read ONLY these two arrays (one branch-free vector load per thread instead of four
   * dependent ones); mean / rstd / gamma / beta are still what the backward kernels use. */
  const float* scale; /* [N*C] or NULL */
  const float* shift; /* [N*C] or NULL */
} mmtta_norm_on_load;

const char* mmtta_last_error(void);   /* thread-local text for the last non-zero status */
int mmtta_abi_version(void);

/* Measurement aid (bench.py's roofline pass): key MMTTA_OPT_PROFILE_MAIN_KERNEL_ONLY != 0 makes the multi-kernel
 * entry points (mmtta_conv_wgrad: main kernel + slab reductions; split-K mmtta_conv_run: main kernel + finalize)
 * launch ONLY their main kernel, so that two events around the call time exactly the kernel rocprofv3 names.
 * Results of such calls are not valid outputs.  Returns the previous value. */
#define MMTTA_OPT_PROFILE_MAIN_KERNEL_ONLY 1
/* Launch-geometry knobs, read when a convolution is PLANNED or LAUNCHED; they change how work is split, never the
 * result beyond fp32 summation order.  A caller that caches mmtta_conv_plan results (workspace size, statistics rows)
 * must drop them after changing a knob and set knobs before capturing launches into a graph (the Python layer does
 * both: ops.set_option).  Defaults are the measured optimum for four volumes in flight per GPU (DESIGN.md section
 * 3.3); `scripts/sweep_tuning.py` sweeps them inside one process.
 *   SPLITK_BELOW / SPLITK_TARGET  implicit GEMM: split the reduction when a launch has fewer workgroups than BELOW, up
 *                                 to about TARGET workgroups                                    (defaults 96 / 128)
 *   WGRAD_WORKGROUPS              workgroups (slabs x channel blocks) of a weight-gradient launch      (default 128)
 *   WGRAD_THIN_SLABS              slabs of the thin-layer weight gradient (<= 4 channels on one side)   (default 256)
 * Returns the previous value, MMTTA_ERR_INVALID for an unknown key or a value < 1. */
#define MMTTA_OPT_SPLITK_BELOW 2
#define MMTTA_OPT_SPLITK_TARGET 3
#define MMTTA_OPT_WGRAD_WORKGROUPS 4
#define MMTTA_OPT_WGRAD_THIN_SLABS 5
/* 1 (default): the bf16 3x3x3 stride-1 stages of the implicit GEMM use the row-structured loader (8-channel items, a
 * thread owns one (x, channel chunk) column of the halo box, geometry paid once per tile, 32-bit offsets; needs strides
 * < 2^24 and < 2^31 elements per tensor, else the generic loader runs) and request the first passes of stage k+1 while
 * the matrix cores work on stage k; 0: the generic loader, load -> barrier -> MFMA -> barrier as in round 1.  Same
 * results bit for bit. */
#define MMTTA_OPT_IGEMM_PIPELINE 6
/* 1 (default): implicit-GEMM epilogues store 16 bytes per lane through an LDS transposition (same values; the
 * statistics rows sum in a different order); 0: four-byte stores straight from the accumulators (round 1). */
#define MMTTA_OPT_EPILOGUE_VEC16 9
/* 1 (default): the 32-output-channel stride-1 layers of bf16 precision (the 64^3 level of the U-Net) use the lean 4x8x8
 * tile (two row blocks per wave: half the accumulator registers, twice the workgroups: 1024 at 64^3, three to four per CU,
 * so that another workgroup's MFMAs cover a workgroup's staging); 0: the 8x8x8 tile (512 workgroups, two per CU).  Same
 * products in the same order: outputs equal bit for bit, the statistics rows are per tile.  Measured: neutral when the
 * implicit GEMM staged one item per trip (r02c), +1.2 % volumes/s with the row loaders (same-box A/B 65.2 against 64.4).
 * Changes the statistics rows a convolution writes: set before planning. */
#define MMTTA_OPT_IGEMM_LEAN 10
/* Kernel of the bf16-operand weight gradients (csrc/conv_wgrad.hip):
 *   1 (default)  the transposed-read kernels for every operand pair that admits their 16-byte items (16-byte-aligned
 *      rows, strides < 2^24, < 2^31 elements): operands stay [voxel][channel] in LDS as in HBM and ds_read_b64_tr_b16
 *      transposes them on the way into the MFMA (27 taps: wgrad_tr_kernel, 1x1x1: wgrad_tr1_kernel);
 *   0  (and every operand pair the above cannot take) the fp32-operand kernel: exact fp32 products.
 * Equal within the bf16 operand rounding (tests/test_hip_conv.py::test_transposed_read_wgrad). */
#define MMTTA_OPT_WGRAD_VECTOR_STAGING 11
/* Stride-2 transposed forms (ConvTranspose3d forward, input gradient of a stride-2 Conv3d) in bf16 mode: when one
 * workgroup per coarse 4 x 4 x 8 tile and 32 output channels makes at least this many workgroups (default 128), all 8
 * output parity classes of a tile are produced by ONE workgroup from one staged halo box (csrc/conv_igemm.hip,
 * igemm_cls8_kernel) instead of 8 x tiles workgroups that each stage their own; 0: never.  Same products, the taps of a
 * class summed before the channel stages instead of after: equal to the per-class kernel within fp32 summation order.
 * Changes the statistics rows a convolution writes (mmtta_conv_plan reports them): set before planning. */
#define MMTTA_OPT_CLASS_FUSED_MIN_WORKGROUPS 12
/* 1 / 2 (default) / 3: in bf16 precision the 3x3x3 stride-1 convolutions with <= 4 channels on both sides (forward and input
 * gradient of the U-Net's last residual unit) run on v_mfma_f32_4x4x4_16B_bf16 (operands rounded to bf16 like every other
 * matrix-core layer of that mode, one instruction per tap and 64 voxels); 0: fp32 FMAs on the vector ALU
 * (direct_row_kernel), as in fp32 precision.  Measured round 2 (3 -> 3 at 128^3): 33 us against 43 us per launch, +0.5 to
 * +1 % volumes/s; full-size parity against the fp32 oracle unchanged (logits 1.496e-2 vs 1.494e-2 of max, Dice 6e-5).
 * The value picks the workgroup tile: 1 = 8 x 8 x 64 voxels (two workgroups per CU), 2 = 4 x 8 x 64 (four per CU: another
 * workgroup's MFMAs cover a workgroup's staging; same-box A/B 64.1 against 63.7 volumes/s), 3 = 2 x 8 x 64.
 * Changes the statistics rows such a convolution writes: set before planning. */
#define MMTTA_OPT_THIN_MFMA 13
int mmtta_set_option(int key, int value);

/* ------------------------------------------------------------------ layout (boundary) ---- */
/* NCDHW fp32 <-> channels-last.  Stands in for nothing in the reference: it is the price of
 * the internal layout, paid once per volume on the way in (reference tensor contract:
 * src/datasets/brats.py:343-347, image float32 [C,D,H,W]) and once on the way out
 * (src/evaluation/seg_eval.py:300, logits [B,R,D,H,W]).  Any strides are accepted; `src` is fp32, `dst` fp32 or bf16
 * (round to nearest even: the staged network input of bf16 precision, 8-byte voxels for <= 4 channels). */
int mmtta_copy_strided(const mmtta_tensor* src, const mmtta_tensor* dst, void* stream);

/* ------------------------------------------------------------------ convolution ---------- */
typedef enum {
  MMTTA_CONV_FWD = 0,    /* torch.nn.Conv3d forward,       y = conv(x, W) + b                */
  MMTTA_CONV_DGRAD = 1,  /* its input gradient,            dx = conv_transpose(dy, W)        */
  MMTTA_CONVT_FWD = 2,   /* torch.nn.ConvTranspose3d fwd   (k3 s2 p1 op1), W is [Cin,Cout,k] */
  MMTTA_CONVT_DGRAD = 3  /* its input gradient,            dx = conv(dy, W), stride 2        */
} mmtta_conv_op;

typedef struct {
  int32_t op;      /* mmtta_conv_op */
  int32_t ksize;   /* 1 or 3 (padding (k-1)/2, dilation 1, groups 1) */
  int32_t stride;  /* 1 or 2 (ConvTranspose: 2 only) */
  int32_t cin;     /* channels of the module's INPUT  (Conv3d.in_channels)  */
  int32_t cout;    /* channels of the module's OUTPUT (Conv3d.out_channels) */
  int32_t dtype;   /* arithmetic: MMTTA_F32 (fp32 MFMA, exact fp32) or MMTTA_BF16 */
} mmtta_conv_desc;

/* Size in bytes of the packed weight image for (desc.op): the MFMA-ready copy
 * [tap][K][N] (K = reduction channels, N = produced channels, both zero padded). */
int64_t mmtta_conv_packed_bytes(const mmtta_conv_desc* desc);

/* Repack master weights (torch layout, fp32: Conv3d [Cout,Cin,k,k,k]; ConvTranspose3d
 * [Cin,Cout,k,k,k]) into the image used by mmtta_conv_run for desc.op.
 * Runs once per optimizer step. */
int mmtta_conv_pack_weights(const mmtta_conv_desc* desc, const float* w_master, void* packed, void* stream);

/* The same for MANY images in one launch (every conv of a model, both orientations): the caller builds the
 * table once on the host (pointers are stable: arena + packed buffers), copies it to the device and then calls
 * mmtta_conv_pack_batched once per optimizer step. */
typedef struct {
  mmtta_conv_desc desc;
  const float* w_master;  /* device */
  void* packed;           /* device */
} mmtta_pack_item;
int64_t mmtta_conv_pack_table_bytes(int count);
/* `total` is an opaque work count produced by _table_build and handed back to _pack_batched. */
int mmtta_conv_pack_table_build(const mmtta_pack_item* items, int count, void* table_host, int64_t* total);
int mmtta_conv_pack_batched(const void* table_dev, int count, int64_t total, void* stream);

/* Launch geometry chosen for a problem; filled by mmtta_conv_plan. */
typedef struct {
  int32_t tiles;         /* M tiles (over n and space) per launch                           */
  int32_t launches;      /* 1, or 8 parity classes for stride-2 transposed forms             */
  int32_t ksplit;        /* >1: partial sums go through the workspace                        */
  int32_t stats_rows;    /* rows of the [rows][2][C] partial-statistics slab this op writes  */
  int32_t config;        /* tile configuration 0..5 (which template instance runs; for profiling) */
  int32_t _pad;
  int64_t workspace_bytes;
} mmtta_conv_plan_t;

/* `x` is the tensor the op READS, `y` the tensor it PRODUCES (for *_DGRAD: x = dy, y = dx). */
int mmtta_conv_plan(const mmtta_conv_desc* desc, const mmtta_tensor* x, const mmtta_tensor* y,
                    mmtta_conv_plan_t* plan);

/* Optional epilogue of mmtta_conv_run: y = acc + bias + T(add), T = norm-on-load of `add`.
 * Fuses MONAI ResidualUnit's `cx + res` (reference: unet_multimodal_midfusion.py:45-55,
 * 121-131 through monai ResidualUnit.forward) into the residual convolution. */
typedef struct {
  const mmtta_tensor* add;          /* NULL: nothing added; same shape as y, channels-last */
  mmtta_norm_on_load add_norm;      /* transform of `add` (mean NULL: added as is)         */
} mmtta_conv_epilogue;

/* One convolution-shaped op as an implicit GEMM on the matrix cores.
 *   Replaces: torch.nn.Conv3d / ConvTranspose3d forward and autograd's input-gradient,
 *   reached from reference src/models/unet.py:68-69 (monai UNet) and
 *   src/models/unet_multimodal_midfusion.py:204-267; backward from
 *   src/core/trainers/seg_trainer.py:142 (loss.backward()).
 *   x        tensor read (channels-last), with an optional norm-on-load
 *   packed   image from mmtta_conv_pack_weights for the same desc
 *   bias     [produced channels] fp32 or NULL
 *   y        tensor written (channels-last); accumulate != 0: y += result
 *   stats    NULL or fp32 [plan.stats_rows][2][C_produced]: per-tile sum / sum of squares of the
 *            values written (input of mmtta_norm_stats_finalize)
 *   workspace  plan.workspace_bytes bytes or NULL when 0 */
int mmtta_conv_run(const mmtta_conv_desc* desc, const mmtta_tensor* x, const mmtta_norm_on_load* x_norm,
                   const void* packed, const float* bias, const mmtta_conv_epilogue* epi,
                   const mmtta_tensor* y, int accumulate, float* stats, void* workspace,
                   int64_t workspace_bytes, void* stream);

/* ---- per-volume parameter sets: N volumes (or N identical sub-networks) in ONE launch, each with ITS OWN parameters.
 * Episodic adaptation has no cross-volume state: every test volume adapts its own copy of the source weights (SURVEY.md
 * Appendix C), and the M modality encoders of the deep-fusion network are M identical graphs with different weights run
 * one after another by the reference (reference src/models/unet_multimodal_midfusion.py:214-218).  Alone, a volume's
 * launches at the 8^3 / 16^3 / 32^3 levels fill a quarter of the chip or less; here they become the batch items of one
 * launch and the batch index selects the parameter set:
 *     set q of batch item n:   q = n / items_per_set
 *     its parameters:          base + (q / inner) * outer + (q % inner) * inner_stride
 * for the packed weight images (`packed_*`, BYTES) and for what lives in the fp32 parameter arenas: weight gradients
 * (`weight_*`, ELEMENTS) and bias vectors / bias gradients (`bias_*`, ELEMENTS; the arenas of a group of volumes are replicas
 * of one layout, so the outer strides are normally all the replica stride, while the inner ones differ).  Examples:
 * G volumes through a U-Net layer: items_per_set 1, inner 1, outer = replica stride.  G volumes x M modality encoders:
 * batch G*M, items_per_set 1, inner M, inner stride = one encoder's parameters.  The fusion layer the reference applies M
 * times with shared weights: batch G*M, items_per_set M, inner 1 (its weight gradient then sums over the M items of a
 * volume).  NULL = one parameter set for the whole batch (the plain entry points).  Strides must keep 16-byte alignment.
 * Every batch item is computed exactly as if it had been launched alone with its set: same tiles, same reduction splits,
 * same summation order - grouped and one-at-a-time runs agree bit for bit (tests/test_hip_groups.py). */
typedef struct {
  int32_t items_per_set; /* consecutive batch items that share a parameter set (>= 1; must divide N) */
  int32_t inner;         /* sets per outer step (>= 1) */
  int64_t packed_outer, packed_inner; /* BYTES between packed weight images */
  int64_t weight_outer, weight_inner; /* ELEMENTS between weight gradients dw (torch weight layout) */
  int64_t bias_outer, bias_inner;     /* ELEMENTS between bias vectors / bias gradients db */
} mmtta_param_sets;

/* mmtta_conv_run with per-item parameter sets: `packed` and `bias` are the base pointers of set 0. */
int mmtta_conv_run_sets(const mmtta_conv_desc* desc, const mmtta_tensor* x, const mmtta_norm_on_load* x_norm,
                        const void* packed, const float* bias, const mmtta_conv_epilogue* epi,
                        const mmtta_tensor* y, int accumulate, float* stats, void* workspace,
                        int64_t workspace_bytes, const mmtta_param_sets* sets, void* stream);

/* Weight (and bias) gradient.
 *   Replaces: autograd's weight-gradient of Conv3d / ConvTranspose3d
 *   (reference src/core/trainers/seg_trainer.py:142).
 *   desc.op  MMTTA_CONV_FWD or MMTTA_CONVT_FWD (which module the gradient is for)
 *   x, x_norm the module's forward input (with the transform it was read with)
 *   dy       gradient of the module's output
 *   dw       fp32, torch layout of the module's weight; accumulate != 0: dw += result
 *   db       fp32 [cout] or NULL */
int64_t mmtta_conv_wgrad_workspace_bytes(const mmtta_conv_desc* desc, const mmtta_tensor* x,
                                         const mmtta_tensor* dy);
/* Which weight-gradient kernel mmtta_conv_wgrad picks for this problem (profiling / roofline bookkeeping):
 *   0 fp32 MFMA k3 s1   1 fp32 MFMA k3 s2 (and conv_transpose)   2 fp32 MFMA k1
 *   3 small-channel (<= 4 channels on one side, fp32 MFMA)       4 bf16 MFMA k3 s1   5 bf16 MFMA k3 s2
 *   6 tiny (<= 4 channels on both sides, k3 s1: fp32 VALU)
 * or a negative mmtta error code. */
int mmtta_conv_wgrad_kernel(const mmtta_conv_desc* desc, const mmtta_tensor* x, const mmtta_tensor* dy);
int mmtta_conv_wgrad(const mmtta_conv_desc* desc, const mmtta_tensor* x, const mmtta_norm_on_load* x_norm,
                     const mmtta_tensor* dy, float* dw, float* db, int accumulate, void* workspace,
                     int64_t workspace_bytes, void* stream);

/* The weight gradient of every parameter set of the batch in one launch sequence: `dw` / `db` are the base pointers of
 * set 0; set q's gradient is the sum over ITS batch items only (slabs never mix sets), reduced in the order the plain call
 * uses for a batch of items_per_set.  Workspace: mmtta_conv_wgrad_workspace_bytes_sets. */
int64_t mmtta_conv_wgrad_workspace_bytes_sets(const mmtta_conv_desc* desc, const mmtta_tensor* x, const mmtta_tensor* dy,
                                              const mmtta_param_sets* sets);
int mmtta_conv_wgrad_sets(const mmtta_conv_desc* desc, const mmtta_tensor* x, const mmtta_norm_on_load* x_norm,
                          const mmtta_tensor* dy, float* dw, float* db, int accumulate, void* workspace,
                          int64_t workspace_bytes, const mmtta_param_sets* sets, void* stream);

/* ------------------------------------------------------------------ input pre-pass -------- */
/* Per-channel intensity rule of one image [C,D,H,W] (reference src/datasets/transforms.py:129-223).
 *   legacy != 0: y = (x - mean) / std                                            (:202-223)
 *   else: optional clip to [lo,hi], then optional z-score with mean / std(unbiased=False, floor eps) taken over the
 *   clipped voxels > mask_gt when `masked` and at least min_count of them exist, over all voxels otherwise (:163-198) */
typedef struct {
  int32_t clip, zscore, masked, min_count;
  float lo, hi, mask_gt, eps;
  float mean, std;
  int32_t legacy, _pad;
} mmtta_intensity_rule;
int64_t mmtta_intensity_scratch_bytes(int channels);
/* x, y: n == 1, NCDHW boundary layout (dense voxels per channel); rules: HOST array of x->c entries. */
int mmtta_intensity_normalize(const mmtta_tensor* x, const mmtta_intensity_rule* rules, const mmtta_tensor* y,
                              void* scratch, void* stream);

/* ------------------------------------------------------------------ normalisation -------- */
typedef enum {
  MMTTA_NORM_INSTANCE = 0, /* torch.nn.InstanceNorm3d(affine=False): statistics per (n,c)      */
  MMTTA_NORM_BATCH = 1,    /* torch.nn.BatchNorm3d: statistics per c over (n,d,h,w)            */
  MMTTA_NORM_GROUP = 2     /* torch.nn.GroupNorm: statistics per (n, group of C/groups chans)  */
} mmtta_norm_kind;

/* Turn per-tile partial sums into the per-(n,c) mean / rstd that consumers read.
 *   Replaces: the statistics half of F.instance_norm / F.batch_norm / F.group_norm (biased
 *   variance, eps inside the sqrt), reached from monai ADN "N".
 *   part     fp32 [rows][2][C] from mmtta_conv_run (rows = rows_per_n * N, n-major)
 *   scratch  fp64 [N*C*2] (fp64 totals between the two stages)
 *   count    voxels per (n,c) = D*H*W
 *   running_mean/var, momentum: BatchNorm only.  training != 0: batch statistics are used and
 *            the running buffers get the EMA update (unbiased variance), the "norm-stat
 *            update" of the adaptation step; training == 0: running statistics are used.
 *   mean, rstd   fp32 [N*C] outputs
 *   gamma, beta  [C] affine parameters or NULL; scale, shift  fp32 [N*C] outputs or NULL:
 *                scale = rstd*gamma, shift = beta - mean*scale (the precombined norm-on-load form) */
int mmtta_norm_stats_finalize(int kind, int groups, const float* part, int rows_per_n, int n, int c,
                              int64_t count, float eps, int training, float* running_mean,
                              float* running_var, float momentum, float* mean, float* rstd,
                              const float* gamma, const float* beta, float* scale, float* shift,
                              double* scratch, void* stream);

/* Rows per batch item of the partial slabs written by mmtta_channel_stats and
 * mmtta_norm_bwd_reduce for a tensor of this shape (deterministic two-stage reductions). */
int mmtta_reduce_rows_per_n(const mmtta_tensor* t);

/* Per-(n,c) partial sum / sum-of-squares of a tensor that no convolution epilogue produced.
 * part: fp32 [N * mmtta_reduce_rows_per_n(x)][2][C]. */
int mmtta_channel_stats(const mmtta_tensor* x, float* part, void* stream);

/* out = Ta(a) + Tb(b)  (b may be NULL), Ta/Tb = norm-on-load.  Materialises
 * relu(norm(y)) [+ residual]: monai ResidualUnit.forward's add with an Identity residual. */
int mmtta_combine(const mmtta_tensor* a, const mmtta_norm_on_load* ta, const mmtta_tensor* b,
                  const mmtta_norm_on_load* tb, const mmtta_tensor* out, void* stream);

/* Backward of norm(+ReLU), pass 1: per-(n,c) reductions
 *   s1[n,c] = sum dz, s2[n,c] = sum dz * xhat,  dz = dout * [post > 0] (relu) or dout,
 *   xhat = (y - mean) * rstd,  post = gamma*xhat + beta.
 *   part: fp32 [N * mmtta_reduce_rows_per_n(y)][2][C].  Replaces the reduction half of
 *   native_*_norm_backward. */
int mmtta_norm_bwd_reduce(const mmtta_tensor* dout, const mmtta_tensor* y, const mmtta_norm_on_load* t,
                          float* part, void* stream);

/* Pass 2 coefficients: from part -> m1, m2 [N*C] (group means of gamma*dz and gamma*dz*xhat) and,
 * if gamma is trained, dgamma / dbeta [C] (accumulate != 0: +=).  training == 0 (BatchNorm on
 * running statistics): m1 = m2 = 0. */
int mmtta_norm_bwd_finalize(int kind, int groups, const float* part, int rows_per_n, int n, int c, int64_t count,
                            const float* gamma, int training, float* m1, float* m2, float* dgamma,
                            float* dbeta, int accumulate, double* scratch /* fp64 [N*C*2] */, void* stream);

/* Pass 3: dy = rstd * (gamma*dz - m1 - xhat*m2); dy may alias dout. */
int mmtta_norm_bwd_apply(const mmtta_tensor* dout, const mmtta_tensor* y, const mmtta_norm_on_load* t,
                         const float* m1, const float* m2, const mmtta_tensor* dy, void* stream);

/* The three passes in ONE launch for a small tensor (the deep levels of the U-Net: launch latency was all their cost):
 * InstanceNorm statistics (the m1 / m2 of mmtta_norm_bwd_finalize with kind = MMTTA_NORM_INSTANCE and count = voxels),
 * no affine gradients, <= 4096 voxels per batch item, C a multiple of 32, voxel-dense 16-byte aligned tensors
 * (mmtta_norm_bwd_small_ok answers 1 / 0 on the host).  One workgroup per 32 channels sums dz and dz*xhat over the voxels
 * (64 partials per channel added in a fixed order in fp64) and writes dy; dy may alias dout.  Same formulas as the
 * three-pass form, another fp32 summation order of the two sums.  Measured (unet 4x128^3, four volumes in flight): +0.5 %
 * when taken for <= 512 voxels (the 8^3 levels), -1.7 % when taken up to 4096 (few workgroups, each streaming its voxels
 * twice): the host side of this repository uses it for <= 512. */
int mmtta_norm_bwd_small_ok(const mmtta_tensor* dout, const mmtta_tensor* y, const mmtta_norm_on_load* t,
                            const mmtta_tensor* dy);
int mmtta_norm_bwd_small(const mmtta_tensor* dout, const mmtta_tensor* y, const mmtta_norm_on_load* t, int64_t count,
                         const mmtta_tensor* dy, void* stream);

/* ------------------------------------------------------------------ resampling / glue ---- */
/* nn.Upsample(scale_factor=2, mode="trilinear", align_corners=True) and its adjoint
 * (reference: src/models/unet_multimodal_midfusion.py:114-120,134 via monai UpSample).  Both tensors of a call share one
 * storage type (fp32 or bf16: activations under method.storage, gradients under method.grad_storage); fp32 arithmetic. */
int mmtta_upsample2x_fwd(const mmtta_tensor* x, const mmtta_tensor* y, void* stream);
int mmtta_upsample2x_bwd(const mmtta_tensor* dy, const mmtta_tensor* dx, int accumulate, void* stream);

/* out = sum_i w[i] * in[i], i < count <= 8 (accumulate != 0: out += ...).  The M-way modality
 * means torch.stack(...).mean (reference: unet_multimodal_midfusion.py:221,229,247), the
 * f_shared + residual add (:96) and their adjoints. */
int mmtta_lincomb(int count, const mmtta_tensor* const* in, const float* w, const mmtta_tensor* out,
                  int accumulate, void* stream);

/* ------------------------------------------------------------------ loss ----------------- */
/* Entropy-minimisation objective and its gradient in one pass (BUILD-DEFINED: the reference
 * has no TTA loss; it takes the place of DiceCELoss in the step
 * skeleton of reference src/core/trainers/seg_trainer.py:141-142).
 *   softmax == 0: mean over (n,r,voxel) of H_bern(z) = softplus(z) - z*sigmoid(z)
 *   softmax != 0: mean over (n,voxel)   of H_cat(z)  = logsumexp_r z - sum_r p_r z_r
 *   logits, dlogits  channels-last, same shape; dlogits = dLoss/dlogits.  logits fp32; dlogits fp32, or - softmax == 0, <= 4
 *            channels in dense 4-channel voxel rows - bf16 (8-byte voxels: the thin gradients of method.grad_storage)
 *   partial  fp64 [mmtta_entropy_partials(...)] scratch; loss  fp32 [1] */
int64_t mmtta_entropy_partials(const mmtta_tensor* logits);
int mmtta_entropy_loss(const mmtta_tensor* logits, int softmax, const mmtta_tensor* dlogits,
                       double* partial, float* loss, void* stream);

/* The same objective for N INDEPENDENT volumes in one launch (a group of volumes adapting side by side, each with its
 * own parameters): loss[n] = mean over batch item n alone, dlogits of item n scaled by item n's count - exactly what N
 * calls of mmtta_entropy_loss on the N items would produce (same block partials, same order).
 *   partial  fp64 [N * mmtta_entropy_partials(one item)] scratch; loss  fp32 [N] */
int64_t mmtta_entropy_partials_items(const mmtta_tensor* logits);
int mmtta_entropy_loss_items(const mmtta_tensor* logits, int softmax, const mmtta_tensor* dlogits,
                             double* partial, float* loss, void* stream);

/* ------------------------------------------------------------------ optimizer ------------ */
/* torch.optim.Adam (amsgrad=False, coupled L2) over a flat parameter arena, two segments:
 * [0, n_decay) with weight_decay, [n_decay, n) without - the decay / no-decay groups of
 * reference src/core/experiment_manager.py:199-237; hyper-parameters
 * configs/training/default.yaml:30-39.  `step` is a device int32 incremented by this call
 * (t starts at 1), so a captured graph replays correctly.  The call that finds *step == 0 starts from ZERO moments whatever
 * m / v hold (torch creates the state as zeros) and does not read them: resetting the optimizer is `*step = 0`, no buffer
 * needs clearing.  SURVEY.md Appendix E K8. */
int mmtta_adam_step(float* p, const float* g, float* m, float* v, int64_t n, int64_t n_decay, float lr,
                    float beta1, float beta2, float eps, float weight_decay, int32_t* step, void* stream);

/* The three optimizers the reference's factory can build (reference src/core/experiment_manager.py:199-210:
 * torch.optim.SGD / Adam / AdamW, chosen by `training.optimizer`, hyper-parameters from
 * `training.optimizers.<name>`, configs/training/default.yaml:11-45) over the same arena layout:
 *   ADAM   as mmtta_adam_step;
 *   ADAMW  p *= 1 - lr*weight_decay (decay segment), then the Adam update without the L2 term;
 *   SGD    g += weight_decay*p (decay segment); buf = g on the first step, momentum*buf + (1-dampening)*g after;
 *          g = nesterov ? g + momentum*buf : buf (momentum 0: plain);  p -= lr*g.   `m` is the momentum buffer,
 *          `v` is unused (may be NULL).
 * All buffers 16-byte aligned, n_decay a multiple of 4 (the arena guarantees both). */
enum { MMTTA_OPTIM_ADAM = 0, MMTTA_OPTIM_ADAMW = 1, MMTTA_OPTIM_SGD = 2 };
typedef struct mmtta_optim_desc {
  int32_t kind;
  float lr, beta1, beta2, eps, weight_decay;
  float momentum, dampening;
  int32_t nesterov;
} mmtta_optim_desc;
int mmtta_optim_step(const mmtta_optim_desc* desc, float* p, const float* g, float* m, float* v, int64_t n,
                     int64_t n_decay, int32_t* step, void* stream);

/* mmtta_optim_step over `sets` replicas of the arena in one launch (a group of volumes, each adapting its own copy):
 * replica r occupies [r * set_stride, r * set_stride + n) of p / g / m / v, its first n_decay elements decay.  One shared
 * step counter, advanced once.  set_stride a multiple of 4. */
int mmtta_optim_step_sets(const mmtta_optim_desc* desc, float* p, const float* g, float* m, float* v, int64_t n,
                          int64_t n_decay, int sets, int64_t set_stride, int32_t* step, void* stream);

/* ------------------------------------------------------------------ evaluation tail ------ */
/* sigmoid -> (>= threshold) -> uint8 mask; GT (> 0.5); per (n,r) integer counts
 * inter = sum p&g, psum = sum p, gsum = sum g.  Replaces reference
 * src/evaluation/seg_eval.py:304-306 and the three reductions of :55-60.
 *   logits  fp32, any strides;  label  any strides, fp32 {0,1};
 *   counts  int64 [N][R][3], zeroed by this call;  mask  uint8 NCDHW [N,R,D,H,W] or NULL */
int mmtta_mask_dice_counts(const mmtta_tensor* logits, const mmtta_tensor* label, float threshold,
                           int64_t* counts, uint8_t* mask, void* stream);

/* Sums behind monai DiceCELoss as the reference builds it - evaluator: sigmoid=True, hard-coded, for
 * ``evaluation.loss.report_loss`` (reference src/evaluation/seg_eval.py:209-220,395-400; SURVEY.md
 * Appendix A.5).  out fp64 [N][R*3+1] (block partials go through `scratch`, mmtta_dice_ce_scratch_bytes; they are
 * summed in a fixed order: reproducible): per region (sum p*y, sum p, sum y) with
 * p = sigmoid(z) (squares of p, y when squared_pred), then the CE numerator: BCE-with-logits with
 * pos_weight = weight[0] when R == 1, soft-label softmax cross entropy with class weights otherwise.
 * The few scalar operations that turn the sums into the loss value are host arithmetic. */
int64_t mmtta_dice_ce_scratch_bytes(const mmtta_tensor* logits);
/* softmax != 0 (a `training.criterion.softmax: true` head, reference src/core/trainers/seg_trainer.py:41-54): the Dice
 * probabilities are softmax over the channels instead of per-channel sigmoids (R > 1; the CE term is softmax CE either way). */
int mmtta_dice_ce_sums(const mmtta_tensor* logits, const mmtta_tensor* label, const float* weight,
                       int squared_pred, int softmax, double* out, void* scratch, void* stream);

/* d(lambda_dice * Dice + lambda_ce * CE)/d(logits) of the same loss (reduction mean), from the sums above (left
 * on the device): the supervised step of reference src/core/trainers/seg_trainer.py:141-142 without autograd.
 * Class weights scale the Dice terms only when more than one Dice channel exists (monai); include_background == 0
 * drops channel 0 from the Dice mean when R > 1. */
int mmtta_dice_ce_grad(const mmtta_tensor* logits, const mmtta_tensor* label, const float* weight, int squared_pred,
                       int softmax, int jaccard, int include_background, float lambda_dice, float lambda_ce, float smooth_nr,
                       float smooth_dr, const double* sums, const mmtta_tensor* dlogits, void* stream);

/* Surface metrics of the evaluation tail: percentile Hausdorff distance and average surface distance per
 * (volume, region).  Replaces the MONAI calls of reference src/evaluation/seg_eval.py:312-340
 * (`HausdorffDistanceMetric(include_background=True, reduction="none", percentile=95, directed=False)` built at
 * :226-234 and `compute_average_surface_distance(..., symmetric=evaluation.surface.asd_symmetric)`), which run
 * scipy on the host.  Edge voxels = mask minus its 6-neighbourhood erosion; distances are exact Euclidean, weighted
 * by `spacing` (host pointer, 3 doubles in D,H,W order: the order the reference hands `evaluation.seg.spacing` to
 * MONAI); the quantile interpolates linearly in float32 like torch.quantile.
 *   pred_mask  uint8 [N,R,D,H,W] dense (the `mask` output of mmtta_mask_dice_counts)
 *   label      fp32 any strides, ground truth = label > 0.5 (reference :306)
 *   hd, asd    fp32 [N*R] on the device.  Both edge sets empty: hd = asd = NaN; exactly one empty: hd = NaN,
 *              asd = +inf (MONAI returns inf distances there); the caller applies the reference's penalty and
 *              sanitising (:342-355).
 *   scratch    mmtta_surface_scratch_bytes(N*R, D, H, W) bytes (negative: extent above 1024 per axis).
 * Results do not depend on scheduling (radix select + integer fixed-point sum): bitwise reproducible. */
int64_t mmtta_surface_scratch_bytes(int64_t n_masks, int64_t d, int64_t h, int64_t w);
int mmtta_surface_distances(const uint8_t* pred_mask, const mmtta_tensor* label, const double* spacing,
                            double percentile, int asd_symmetric, float* hd, float* asd, void* scratch, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MMTTA_H */
