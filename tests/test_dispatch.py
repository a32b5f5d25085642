"""Every dispatchable kernel instantiation of the convolution family is selected by at least one shape of the GPU
parity cases (tests/test_hip_conv.py::CASES / BF16_CASES) - asserted through the planner the launches themselves use
(``mmtta_conv_plan`` / ``mmtta_conv_wgrad_kernel`` are host-only, so this runs without a GPU too).

config ids: csrc/conv_igemm.hip::config_id (0-5 fp32 igemm, 6 direct, 7-12 bf16 igemm, 13 channel kernel, 14 the lean 4x8x8 tile
of the 32-output-channel bf16 layers: the default, id 7 = the 8x8x8 tile under MMTTA_OPT_IGEMM_LEAN = 0, parity case
test_lean_tile_matches_the_wide_tile; 15 class-fused);
weight-gradient ids: mmtta_conv_wgrad_kernel (0 f32 s1, 1 f32 s2, 2 1x1, 3 small-channel, 6 tiny, 7 / 8 transposed-read
bf16 s1 / s2: the default of bf16 precision, parity cases test_transposed_read_wgrad; 0 / 1 under option 11 = 0; ids 4 / 5
were round 1's staging-transposed bf16 kernels, removed in round 3;
9 the 1x1x1 streaming kernel of bf16 precision, parity cases test_pointwise_conv_weight_gradient_on_transposed_reads;
10 the thin 27-tap layers of bf16 precision on transposed reads, parity cases test_thin_wgrad_on_transposed_reads).
VERDICT r1 P1: ids 2 and 9 (igemm <4,4,4,4,8,32>) used to be reachable only by the full-size bench."""
import ctypes as C

import torch

from test_hip_conv import BF16_CASES, CASES, THIN_TR_CASES, TR_CASES

# (case, (fwd, dgrad) config in fp32 mode, the same in bf16 mode, weight-gradient kernel (fp32 mode, bf16 mode))
DISPATCH = [
    ((32, 32, 3, 1, False, (1, 16, 16, 16)), (0, 0), (14, 14), (0, 7)),
    ((64, 64, 3, 1, False, (1, 8, 8, 8)), (1, 1), (8, 8), (0, 7)),
    ((128, 128, 3, 1, False, (1, 32, 32, 32)), (2, 2), (9, 9), (0, 7)),
    ((16, 32, 3, 2, False, (1, 8, 8, 8)), (3, 0), (10, 14), (1, 8)),
    ((32, 64, 3, 2, False, (1, 8, 8, 16)), (4, 0), (11, 14), (1, 8)),
    ((64, 128, 3, 2, False, (1, 5, 6, 7)), (5, 1), (12, 8), (1, 8)),
    ((128, 136, 3, 1, False, (2, 4, 6, 8)), (5, 5), (12, 12), (0, 7)),
    ((768, 128, 3, 2, True, (1, 8, 8, 8)), (2, 5), (9, 12), (1, 8)),
    ((256, 512, 1, 1, False, (1, 4, 4, 4)), (5, 5), (12, 12), (2, 9)),
    ((64, 3, 3, 2, True, (1, 3, 5, 70)), (6, 4), (6, 13), (3, 10)),
    ((4, 32, 3, 2, False, (1, 16, 16, 16)), (13, 6), (13, 6), (3, 10)),
    ((3, 3, 3, 1, False, (1, 8, 8, 8)), (6, 6), (6, 6), (6, 10)),
]


def _desc(t, dtype=None):
    from multimodal_tta_amd._lib import F32, Tensor
    n, d, h, w, c = t.shape
    sn, sd, sh, sw, _ = t.stride()
    return Tensor(t.data_ptr(), n, c, d, h, w, sn, 1, sd, sh, sw, F32 if dtype is None else dtype, 0)


def _cl(n, d, h, w, c):
    return torch.empty((n, d, h, w, (c + 3) // 4 * 4))[..., :c]


def test_every_kernel_instantiation_is_reached_by_a_parity_case():
    from multimodal_tta_amd import _lib
    from multimodal_tta_amd._lib import BF16, CONV_DGRAD, CONV_FWD, CONVT_DGRAD, CONVT_FWD, F32, ConvDesc, ConvPlan

    lib = _lib.load()
    # the launch-geometry knobs are process-wide and an adaptation plugin re-tunes them for its volumes in flight: the table
    # below holds for the documented defaults (4 volumes in flight), whatever ran before in this process
    from multimodal_tta_amd import ops
    ops.tune_for_volumes_in_flight(4)
    seen_cfg, seen_wg, splitk_wide = set(), set(), set()
    for case, want_f32, want_bf16, want_wg in DISPATCH:
        cin, cout, k, stride, transposed, (n, d, h, w) = case
        assert case in CASES, case
        x = _cl(n, d, h, w, cin)
        if transposed:
            y = _cl(n, 2 * d, 2 * h, 2 * w, cout)
        elif stride == 1:
            y = _cl(n, d, h, w, cout)
        else:
            y = _cl(n, (d + 1) // 2, (h + 1) // 2, (w + 1) // 2, cout)
        tx, ty = _desc(x), _desc(y)
        fo, do = (CONVT_FWD, CONVT_DGRAD) if transposed else (CONV_FWD, CONV_DGRAD)
        for dtype, want, wg in ((F32, want_f32, want_wg[0]), (BF16, want_bf16, want_wg[1])):
            if dtype == BF16 and (want != want_f32 or wg != want_wg[0]):
                assert case in BF16_CASES, f"{case}: bf16 instantiation without a bf16 parity case"
            got = []
            for op, a, b in ((fo, tx, ty), (do, ty, tx)):
                dsc, plan = ConvDesc(op, k, stride, cin, cout, dtype), ConvPlan()
                assert lib.mmtta_conv_plan(C.byref(dsc), C.byref(a), C.byref(b), C.byref(plan)) == 0
                got.append(int(plan.config))
                if plan.ksplit > 1:
                    splitk_wide.add(int(plan.config))
            assert tuple(got) == want, f"{case} dtype {dtype}: plan configs {got}, expected {want}"
            dsc = ConvDesc(fo, k, stride, cin, cout, dtype)
            kid = int(lib.mmtta_conv_wgrad_kernel(C.byref(dsc), C.byref(tx), C.byref(ty)))
            assert kid == wg, f"{case} dtype {dtype}: weight-gradient kernel {kid}, expected {wg}"
            seen_cfg.update(got)
            seen_wg.add(kid)
            if dtype == BF16 and 14 in got:                # the 8x8x8 tile (id 7) stays reachable: MMTTA_OPT_IGEMM_LEAN = 0
                prev = lib.mmtta_set_option(10, 0)
                try:
                    for op, a, b, g in ((fo, tx, ty, got[0]), (do, ty, tx, got[1])):
                        dsc0, plan0 = ConvDesc(op, k, stride, cin, cout, dtype), ConvPlan()
                        assert lib.mmtta_conv_plan(C.byref(dsc0), C.byref(a), C.byref(b), C.byref(plan0)) == 0
                        assert int(plan0.config) == (7 if g == 14 else g), f"{case}: option 10 = 0 plans {int(plan0.config)}"
                        seen_cfg.add(int(plan0.config))
                finally:
                    lib.mmtta_set_option(10, prev)
            if dtype == BF16 and wg == 10:
                assert case in THIN_TR_CASES, f"{case}: thin transposed-read instantiation without a parity case"
            if dtype == BF16 and wg in (7, 8):
                assert case in TR_CASES, f"{case}: transposed-read instantiation without a parity case"
                # the same layer with its module input bf16-stored (method.storage: bf16): the same kernel
                kid = int(lib.mmtta_conv_wgrad_kernel(C.byref(dsc), C.byref(_desc(x, BF16)), C.byref(ty)))
                assert kid == wg, f"{case} bf16-stored: weight-gradient kernel {kid}, expected {wg}"
                # MMTTA_OPT_WGRAD_VECTOR_STAGING = 0 (and every operand pair the 16-byte loader cannot take): the fp32-operand
                # kernels (ids 0 / 1)
                prev = lib.mmtta_set_option(11, 0)
                try:
                    kid0 = int(lib.mmtta_conv_wgrad_kernel(C.byref(dsc), C.byref(tx), C.byref(ty)))
                finally:
                    lib.mmtta_set_option(11, prev)
                assert kid0 == wg - 7, f"{case} option 0: weight-gradient kernel {kid0}, expected {wg - 7}"
                seen_wg.add(kid0)
    assert seen_cfg == set(range(15)), f"conv configs without a parity case: {sorted(set(range(15)) - seen_cfg)}"
    live_wg = set(range(11)) - {4, 5}            # ids 4 / 5 were the round-1 staging-transposed bf16 kernels (removed in round 3)
    assert seen_wg == live_wg, f"weight-gradient kernels without a parity case: {sorted(live_wg - seen_wg)}"
    assert {2, 9} <= splitk_wide, "the 32-channel-stage igemm also needs a split-K parity case"


def test_class_fused_kernel_is_planned_for_large_stride2_forms():
    """config 15 (igemm_cls8_kernel): the stride-2 transposed forms of a bf16 layer take the class-fused kernel once one
    workgroup per coarse 4x4x8 tile and 32 output channels makes >= MMTTA_OPT_CLASS_FUSED_MIN_WORKGROUPS (128) workgroups,
    and the per-class kernel below that; its parity cases force it on small shapes
    (tests/test_hip_conv.py::test_class_fused_stride2_forms_match_the_per_class_kernel)."""
    from multimodal_tta_amd import _lib
    from multimodal_tta_amd._lib import BF16, CONV_DGRAD, CONVT_FWD, ConvDesc, ConvPlan

    lib = _lib.load()
    for (cin, cout, transposed, coarse, want) in ((128, 32, True, (32, 32, 32), 15), (32, 64, False, (32, 32, 32), 15),
                                                  (128, 32, True, (8, 8, 8), 14), (768, 128, True, (8, 8, 8), 9)):
        d, h, w = coarse
        lo, hi = _cl(1, d, h, w, cin if transposed else cout), _cl(1, 2 * d, 2 * h, 2 * w, cout if transposed else cin)
        op = CONVT_FWD if transposed else CONV_DGRAD          # both read the coarse tensor and write the fine one
        dsc, plan = ConvDesc(op, 3, 2, cin, cout, BF16), ConvPlan()
        assert lib.mmtta_conv_plan(C.byref(dsc), C.byref(_desc(lo)), C.byref(_desc(hi)), C.byref(plan)) == 0
        assert int(plan.config) == want, (cin, cout, transposed, coarse, int(plan.config))
        if want == 15:
            assert plan.ksplit == 1 and plan.stats_rows == (d // 4) * (h // 4) * (w // 8)
