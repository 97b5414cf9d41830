// Every -D switch the library's sources react to, in ONE place (VERDICT r4 hygiene): what each is, its default, and the string
// bn_build_flags() reports.  The product library is built with none of them (tests/test_host_cpu.py asserts an empty string);
// variant libraries come from `python -m brdf_nerf_amd.build -DSWITCH[=value] --tag=...` and are loaded with
// BRDFNERF_HIP_LIB=... by profiles/ab_kernels.py (variants alternating in one process) and the timing tools.
// Included first by common.h: a switch is RECORDED here before its default is applied, so the report shows what the command
// line set, not what the defaults define.
//
//   kind T (timing / diagnostic code compiled in, results unchanged, slower):
//     BN_PHASE_TIMING, BN_PHASE_TIMING_WGRAD   per-phase cycle counters of the chain kernels / wgrad256 (profiles/phase_timing.py)
//     BN_CLOCK_STAMP, BN_CLOCK_STAMP_WGRAD     s_memtime / s_memrealtime stamps: the clock the chip holds (profiles/clock_probe.py)
//     BN_TIMELINE                               per-wave event log of the forward trunk (profiles/simd_timeline.py)
//   kind A (A/B switch, results unchanged):
//     BN_GEMM_PRIO=<n>       s_setprio of a wave while it multiplies in the barrier-free trunks (default 1)
//     BN_PRIO_YOUNG          static priority for the later-dispatched half of the forward's waves
//     BN_NO_NT_STASH         plain instead of non-temporal stash stores / loads
//     BN_NO_PINGPONG, BN_BWD_NO_PINGPONG        the trunks under workgroup barriers (rounds 1-3) instead of LDS hand-overs
//     BN_PP_LOOP_NKS         the trunks' half-GEMMs as loops with tail steps (rounds 1-4) instead of straight-line code
//     BN_PP_SPLIT            the trunks' layer GEMM as two half-GEMMs with a weight prologue each (rounds 1-4) instead of one stream
//     BN_NO_FIXED_FULL       head passes / sigma head / the backward's top layer in the looped form (round 4) instead of straight-line
//     BN_ADJ_LOOP            the adjoint chain's trunk product in the looped form (round 4) instead of a straight-line buffer-load stream
//     BN_NO_BUFW             the trunks' weight fragments by global loads with vector addresses (rounds 1-4) instead of buffer loads
//     BN_FWD_DEPTH_TRAIN=<n> weight-fragment prefetch depth of the training forward (default 6)
//     BN_BWD_DEPTH=<n>       ... of the backward / adjoint chains under barriers (default 2)
//     BN_BWD_PP_DEPTH=<n>    ... of the barrier-free backward trunk (default 6)
//     BN_BWD_D_AT=<0|1|2>    where the barrier-free backward trunk issues a layer's derivative loads (default 0: before the GEMM)
//     BN_HEAD_WIDE           single-head passes on half the waves with twice the columns each
//     BN_NO_FLAT_COMPOSITE   the per-(sample, channel) scalar compositing path everywhere
//     BN_DPH=<n>             pre-activation gradients kept per point for the heads (default 3 * BN_MAX_HEADS)
//     SKINNY_SPLITS=<n>      point splits of skinny_wgrad_kernel (default 256)
//     BN_W2_BLOCKS=<n>       wgrad256: tiles x point splits per round of the 256 CUs (default 256)
//   kind D (numerical diagnostic, results CHANGED on purpose):
//     BN_DIAG_D8_IN_F32      the fp32 mode sends its activation derivatives through the 16-bit modes' 8-bit codec (Siren layers):
//                            what the 8-bit D stash alone does to the analytic normals (profiles/diag_c5_rows.py --d8lib=...)
//   kind P (timing probe, RESULTS WRONG - never ship):
//     BN_PROBE_NO_A, BN_PROBE_NO_B   chain GEMM without its weight / LDS fragment traffic (profiles/probe_gemm_rate.py)
//     BN_PROBE_NO_D                  backward chain without its derivative loads
//     BN_PROBE_NO_RIDE               row-major stash copy without its global stores
//     BN_ABLATION_BUILD              marker set by profiles/ scripts that patch sources for an ablation
#pragma once

#ifdef BN_PHASE_TIMING
#define BN_F_PHASE_TIMING "BN_PHASE_TIMING "
#else
#define BN_F_PHASE_TIMING ""
#endif
#ifdef BN_PHASE_TIMING_WGRAD
#define BN_F_PHASE_TIMING_WGRAD "BN_PHASE_TIMING_WGRAD "
#else
#define BN_F_PHASE_TIMING_WGRAD ""
#endif
#ifdef BN_CLOCK_STAMP
#define BN_F_CLOCK_STAMP "BN_CLOCK_STAMP "
#else
#define BN_F_CLOCK_STAMP ""
#endif
#ifdef BN_CLOCK_STAMP_WGRAD
#define BN_F_CLOCK_STAMP_WGRAD "BN_CLOCK_STAMP_WGRAD "
#else
#define BN_F_CLOCK_STAMP_WGRAD ""
#endif
#ifdef BN_TIMELINE
#define BN_F_TIMELINE "BN_TIMELINE "
#else
#define BN_F_TIMELINE ""
#endif
#ifdef BN_GEMM_PRIO
#define BN_F_GEMM_PRIO "BN_GEMM_PRIO "
#else
#define BN_F_GEMM_PRIO ""
#define BN_GEMM_PRIO 1
#endif
#ifdef BN_PRIO_YOUNG
#define BN_F_PRIO_YOUNG "BN_PRIO_YOUNG "
#else
#define BN_F_PRIO_YOUNG ""
#endif
#ifdef BN_NO_NT_STASH
#define BN_F_NO_NT_STASH "BN_NO_NT_STASH "
#else
#define BN_F_NO_NT_STASH ""
#endif
#ifdef BN_NO_PINGPONG
#define BN_F_NO_PINGPONG "BN_NO_PINGPONG "
#else
#define BN_F_NO_PINGPONG ""
#endif
#ifdef BN_BWD_NO_PINGPONG
#define BN_F_BWD_NO_PINGPONG "BN_BWD_NO_PINGPONG "
#else
#define BN_F_BWD_NO_PINGPONG ""
#endif
#ifdef BN_PP_LOOP_NKS
#define BN_F_PP_LOOP_NKS "BN_PP_LOOP_NKS "
#else
#define BN_F_PP_LOOP_NKS ""
#endif
#ifdef BN_PP_SPLIT
#define BN_F_PP_SPLIT "BN_PP_SPLIT "
#else
#define BN_F_PP_SPLIT ""
#endif
#ifdef BN_NO_FIXED_FULL
#define BN_F_NO_FIXED_FULL "BN_NO_FIXED_FULL "
#else
#define BN_F_NO_FIXED_FULL ""
#endif
#ifdef BN_ADJ_LOOP
#define BN_F_ADJ_LOOP "BN_ADJ_LOOP "
#else
#define BN_F_ADJ_LOOP ""
#endif
#ifdef BN_NO_BUFW
#define BN_F_NO_BUFW "BN_NO_BUFW "
#else
#define BN_F_NO_BUFW ""
#endif
#ifdef BN_FWD_DEPTH_TRAIN
#define BN_F_FWD_DEPTH_TRAIN "BN_FWD_DEPTH_TRAIN "
#else
#define BN_F_FWD_DEPTH_TRAIN ""
#define BN_FWD_DEPTH_TRAIN 6
#endif
#ifdef BN_BWD_DEPTH
#define BN_F_BWD_DEPTH "BN_BWD_DEPTH "
#else
#define BN_F_BWD_DEPTH ""
#define BN_BWD_DEPTH 2
#endif
#ifdef BN_BWD_PP_DEPTH
#define BN_F_BWD_PP_DEPTH "BN_BWD_PP_DEPTH "
#else
#define BN_F_BWD_PP_DEPTH ""
#define BN_BWD_PP_DEPTH 6
#endif
#ifdef BN_BWD_D_AT
#define BN_F_BWD_D_AT "BN_BWD_D_AT "
#else
#define BN_F_BWD_D_AT ""
#define BN_BWD_D_AT 0
#endif
#ifdef BN_HEAD_WIDE
#define BN_F_HEAD_WIDE "BN_HEAD_WIDE "
#else
#define BN_F_HEAD_WIDE ""
#endif
#ifdef BN_NO_FLAT_COMPOSITE
#define BN_F_NO_FLAT_COMPOSITE "BN_NO_FLAT_COMPOSITE "
#else
#define BN_F_NO_FLAT_COMPOSITE ""
#endif
#ifdef BN_DPH
#define BN_F_DPH "BN_DPH "
#else
#define BN_F_DPH ""
#endif
#ifdef SKINNY_SPLITS
#define BN_F_SKINNY_SPLITS "SKINNY_SPLITS "
#else
#define BN_F_SKINNY_SPLITS ""
#define SKINNY_SPLITS 256   // 512: 0.129 ms, 256: 0.102 ms, 128: 0.169 ms per launch (round 2)
#endif
#ifdef BN_W2_BLOCKS
#define BN_F_W2_BLOCKS "BN_W2_BLOCKS "
#else
#define BN_F_W2_BLOCKS ""
#define BN_W2_BLOCKS 256
#endif
#ifdef BN_DIAG_D8_IN_F32
#define BN_F_DIAG_D8_IN_F32 "BN_DIAG_D8_IN_F32 "
#else
#define BN_F_DIAG_D8_IN_F32 ""
#endif
#ifdef BN_PROBE_NO_A
#define BN_F_PROBE_NO_A "BN_PROBE_NO_A "
#else
#define BN_F_PROBE_NO_A ""
#endif
#ifdef BN_PROBE_NO_B
#define BN_F_PROBE_NO_B "BN_PROBE_NO_B "
#else
#define BN_F_PROBE_NO_B ""
#endif
#ifdef BN_PROBE_NO_D
#define BN_F_PROBE_NO_D "BN_PROBE_NO_D "
#else
#define BN_F_PROBE_NO_D ""
#endif
#ifdef BN_PROBE_NO_RIDE
#define BN_F_PROBE_NO_RIDE "BN_PROBE_NO_RIDE "
#else
#define BN_F_PROBE_NO_RIDE ""
#endif
#ifdef BN_ABLATION_BUILD
#define BN_F_ABLATION_BUILD "BN_ABLATION_BUILD "
#else
#define BN_F_ABLATION_BUILD ""
#endif

// what bn_build_flags() returns (error.cpp): the switches set on the command line of THIS translation unit - variant builds pass
// the same defines to every file
#define BN_BUILD_FLAGS_STRING                                                                                              \
  BN_F_PHASE_TIMING BN_F_PHASE_TIMING_WGRAD BN_F_CLOCK_STAMP BN_F_CLOCK_STAMP_WGRAD BN_F_TIMELINE BN_F_GEMM_PRIO          \
  BN_F_PRIO_YOUNG BN_F_NO_NT_STASH BN_F_NO_PINGPONG BN_F_BWD_NO_PINGPONG BN_F_PP_LOOP_NKS BN_F_PP_SPLIT BN_F_NO_FIXED_FULL BN_F_ADJ_LOOP BN_F_NO_BUFW BN_F_FWD_DEPTH_TRAIN            \
  BN_F_BWD_DEPTH BN_F_BWD_PP_DEPTH BN_F_BWD_D_AT BN_F_HEAD_WIDE BN_F_NO_FLAT_COMPOSITE BN_F_DPH BN_F_SKINNY_SPLITS        \
  BN_F_W2_BLOCKS BN_F_DIAG_D8_IN_F32 BN_F_PROBE_NO_A BN_F_PROBE_NO_B BN_F_PROBE_NO_D BN_F_PROBE_NO_RIDE BN_F_ABLATION_BUILD
