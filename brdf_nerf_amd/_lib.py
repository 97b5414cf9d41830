"""ctypes binding of the C ABI declared in include/brdfnerf_hip.h.

There is NO fallback: if libbrdfnerf_hip.so is missing or fails to load, importing the product
path raises.  (Build it with `python -m brdf_nerf_amd.build` or `__graft_entry__.build()`.)
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# BRDFNERF_HIP_LIB selects another build of the same library (variant builds for A/B measurements, brdf_nerf_amd.build -D...)
LIB_PATH = os.environ.get("BRDFNERF_HIP_LIB") or os.path.join(HERE, "libbrdfnerf_hip.so")

BN_MAX_LAYERS = 12
BN_MAX_HEADS = 6
BN_F32, BN_BF16, BN_F16 = 0, 1, 2
DTYPES = {"fp32": BN_F32, "bf16": BN_BF16, "fp16": BN_F16}
BN_ACT_SIN, BN_ACT_RELU = 0, 1
BN_HEAD_PLAIN, BN_HEAD_RPV_K, BN_HEAD_RPV_THETA, BN_HEAD_HAPKE_THETA, BN_HEAD_TILE3, BN_HEAD_BETA = 0, 1, 2, 3, 4, 5
BN_BRDF_AUX = 16

fptr = C.c_void_p


class FieldDesc(C.Structure):
    _fields_ = [("feat", C.c_int32), ("layers", C.c_int32), ("skip", C.c_int32), ("pe_freqs", C.c_int32),
                ("act", C.c_int32), ("dtype", C.c_int32), ("n_heads", C.c_int32),
                ("head_out", C.c_int32 * BN_MAX_HEADS), ("head_kind", C.c_int32 * BN_MAX_HEADS),
                ("normal_lr", C.c_int32), ("normal_an", C.c_int32), ("out_channels", C.c_int32), ("fold_feats", C.c_int32),
                ("dir_dim", C.c_int32), ("dir_freqs", C.c_int32), ("t_dim", C.c_int32)]


_PARAM_FIELDS = [("trunk_w", fptr * BN_MAX_LAYERS), ("trunk_b", fptr * BN_MAX_LAYERS),
                 ("sigma_w", fptr), ("sigma_b", fptr), ("feats_w", fptr), ("feats_b", fptr),
                 ("head_w1", fptr * BN_MAX_HEADS), ("head_b1", fptr * BN_MAX_HEADS),
                 ("head_w2", fptr * BN_MAX_HEADS), ("head_b2", fptr * BN_MAX_HEADS),
                 ("normal_w", fptr), ("normal_b", fptr), ("head0_wdir", fptr), ("head0_wdir_ld", C.c_int64),
                 ("head1_wt", fptr), ("head1_wt_ld", C.c_int64)]


class FieldParams(C.Structure):
    _fields_ = list(_PARAM_FIELDS)


class FieldGrads(C.Structure):      # the same members, writable, + the gradient of the embedding input
    _fields_ = list(_PARAM_FIELDS) + [("d_t_embed", fptr)]


class Points(C.Structure):
    _fields_ = [("xyz", fptr), ("rays", fptr), ("z", fptr), ("ray_stride", C.c_int32), ("n_samples", C.c_int32),
                ("n_points", C.c_int64), ("dirs", fptr), ("t_embed", fptr), ("point_offset", C.c_int64), ("total_points", C.c_int64),
                ("z2", fptr), ("n_samples2", C.c_int32), ("seg1_points", C.c_int64)]


class ShadeDesc(C.Structure):         # bn_shade_desc
    _fields_ = [(k, C.c_int32) for k in ("kind", "C", "ch_normal", "ch_p0", "ch_p1", "ch_p2", "rhoc_is_albedo", "shell",
                                         "cos_irradiance", "usealldepth")] + \
               [(k, C.c_float) for k in ("hpk_scl", "f0", "rgb_padding", "lambda_rgb", "lambda_ds", "lambda_hs")] + \
               [("irr", fptr), ("irr_stride", C.c_int64)]


BN_SHADE_LAMBERT, BN_SHADE_RPV, BN_SHADE_HAPKE, BN_SHADE_MICROFACET = 0, 1, 2, 3


class NormalReg(C.Structure):         # bn_normal_reg
    _fields_ = [("rays_d", fptr), ("rd_stride", C.c_int64), ("ch_an", C.c_int32), ("ch_lr", C.c_int32),
                ("lambda_an", C.c_float), ("lambda_lr", C.c_float), ("lambda_spv", C.c_float), ("spv_ch_an", C.c_int32),
                ("spv_ch_lr", C.c_int32), ("spv_ray", fptr), ("spv_tot", fptr)]


class Noise(C.Structure):             # bn_noise
    _fields_ = [("rng", fptr), ("noise_std", C.c_float), ("rng_stream", C.c_uint32), ("ray_offset", C.c_int64)]


class FoldDesc(C.Structure):          # bn_fold_desc
    _fields_ = [("n_heads", C.c_int32), ("F", C.c_int32), ("rows", C.c_int32), ("wf", fptr), ("bf", fptr),
                ("w1", fptr * BN_MAX_HEADS), ("w1_ld", C.c_int64 * BN_MAX_HEADS), ("b1", fptr * BN_MAX_HEADS),
                ("w_fold", fptr * BN_MAX_HEADS), ("b_fold", fptr * BN_MAX_HEADS), ("m", fptr * BN_MAX_HEADS), ("s", fptr * BN_MAX_HEADS),
                ("d_w1", fptr * BN_MAX_HEADS), ("d_w1_ld", C.c_int64 * BN_MAX_HEADS), ("d_b1", fptr * BN_MAX_HEADS),
                ("d_wf", fptr), ("d_bf", fptr)]


BN_ABI_VERSION = 7
BN_STATE_BYTES, BN_STATE_LOSS_OFF, BN_STATE_LOSS_SLOTS, BN_STATE_POW_OFF, BN_STATE_PART_OFF = 1024, 64, 64, 320, 512
BN_STATE_NOISE_OFF = 40
BN_RNG_COARSE, BN_RNG_GUIDED, BN_RNG_GUIDED_TARGET, BN_RNG_NOISE_COARSE, BN_RNG_NOISE_MERGED, BN_RNG_SUN = 1, 2, 3, 4, 5, 6
BN_BWD_CHAIN, BN_BWD_WGRAD_TRUNK, BN_BWD_WGRAD_HEADS, BN_BWD_SKINNY, BN_BWD_ALL = 1, 2, 4, 8, 15


class LibraryMissing(RuntimeError):
    pass


_lib = None

_SIGS = {
    "bn_abi_version": (C.c_int, []),
    "bn_source_hash": (C.c_char_p, []),
    "bn_last_error": (C.c_char_p, []),
    "bn_build_flags": (C.c_char_p, []),
    "bn_set_deterministic": (C.c_int, [C.c_int]),
    "bn_get_deterministic": (C.c_int, []),
    "bn_field_packed_bytes": (C.c_size_t, [C.POINTER(FieldDesc)]),
    "bn_pack_field": (C.c_int, [C.POINTER(FieldDesc), C.POINTER(FieldParams), fptr, fptr]),
    "bn_field_stash_bytes": (C.c_size_t, [C.POINTER(FieldDesc), C.c_int64]),
    "bn_field_sigma": (C.c_int, [C.POINTER(FieldDesc), C.POINTER(FieldParams), fptr, C.POINTER(Points), fptr, fptr]),
    "bn_field_forward": (C.c_int, [C.POINTER(FieldDesc), C.POINTER(FieldParams), fptr, C.POINTER(Points), fptr, fptr,
                                   fptr]),
    "bn_field_backward": (C.c_int, [C.POINTER(FieldDesc), C.POINTER(FieldParams), fptr, C.POINTER(Points), fptr, fptr,
                                    fptr, C.POINTER(FieldGrads), fptr]),
    "bn_field_backward_parts": (C.c_int, [C.POINTER(FieldDesc), C.POINTER(FieldParams), fptr, C.POINTER(Points), fptr, fptr,
                                          fptr, C.POINTER(FieldGrads), C.c_int32, fptr]),
    "bn_field_normals": (C.c_int, [C.POINTER(FieldDesc), C.POINTER(FieldParams), fptr, C.POINTER(Points), fptr, fptr, fptr,
                                   C.c_int32, fptr]),
    "bn_composite_forward": (C.c_int, [fptr, fptr, C.c_int64, fptr, C.c_float, fptr, C.c_int64, C.c_int32, C.c_int64,
                                       C.c_int32, fptr, fptr, fptr, fptr, fptr, fptr]),
    "bn_composite_backward": (C.c_int, [fptr, fptr, C.c_int64, fptr, C.c_float, fptr, C.c_int64, C.c_int32, C.c_int64,
                                        C.c_int32, fptr, fptr, fptr, fptr, C.c_int64, fptr, C.c_int64, fptr]),
    "bn_lambert_loss": (C.c_int, [fptr, C.c_int32, fptr, fptr, C.c_int32, fptr, fptr, fptr, fptr, fptr, fptr, C.c_float, C.c_float,
                                  C.c_float, C.c_int32, C.c_int64, fptr, fptr, fptr, fptr, fptr, fptr]),
    "bn_stratified_z": (C.c_int, [fptr, fptr, C.c_int64, fptr, C.c_int64, C.c_int32, fptr, fptr]),
    "bn_guided_samples": (C.c_int, [fptr, fptr, fptr, fptr, C.c_int64, C.c_int32, C.c_int32, C.c_float, C.c_float,
                                    C.c_float, fptr, fptr, fptr, fptr, fptr, fptr, fptr, fptr, fptr]),
    "bn_guided_samples_nf": (C.c_int, [fptr, fptr, fptr, fptr, C.c_int64, C.c_int32, C.c_int32, fptr, C.c_float, fptr, fptr,
                                       fptr, fptr, fptr, fptr, fptr, fptr, fptr]),
    "bn_brdf_rpv_forward": (C.c_int, [fptr] * 7 + [C.c_int64, fptr, fptr, fptr]),
    "bn_brdf_rpv_backward": (C.c_int, [fptr] * 8 + [C.c_int64] + [fptr] * 6),
    "bn_brdf_hapke_forward": (C.c_int, [fptr] * 7 + [C.c_float, C.c_int32, C.c_int64, fptr, fptr, fptr]),
    "bn_brdf_hapke_backward": (C.c_int, [fptr] * 7 + [C.c_float, C.c_int32, fptr, C.c_int64] + [fptr] * 6),
    "bn_brdf_microfacet_forward": (C.c_int, [fptr] * 5 + [C.c_float, C.c_int64, fptr, fptr, fptr]),
    "bn_brdf_microfacet_backward": (C.c_int, [fptr] * 5 + [C.c_float, fptr, C.c_int64] + [fptr] * 4),
    "bn_adam_step": (C.c_int, [fptr, fptr, fptr, fptr, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float,
                               C.c_float, C.c_int32, C.c_float, fptr]),
    "bn_count_nonfinite": (C.c_int, [fptr, C.c_int64, fptr, fptr]),
    "bn_stratified_z_rng": (C.c_int, [fptr, fptr, C.c_int64, fptr, C.c_uint32, C.c_int64, C.c_int64, C.c_int32, fptr, fptr]),
    "bn_rng_uniform": (C.c_int, [fptr, C.c_uint32, C.c_int64, fptr, fptr]),
    "bn_rng_normal": (C.c_int, [fptr, C.c_uint32, C.c_int64, fptr, fptr]),
    "bn_normal_spv_reduce": (C.c_int, [fptr, C.c_int64, C.c_int32, C.c_float, fptr, fptr, fptr, fptr]),
    "bn_composite_guided": (C.c_int, [fptr, fptr, C.c_int64, C.c_int64, C.c_int32, C.c_int32, fptr, C.c_float, fptr, C.c_int64,
                                      fptr, C.c_int64, fptr, C.c_int64, fptr, fptr, fptr, C.c_uint32, C.c_uint32, C.c_int64, fptr, fptr,
                                      fptr, fptr, fptr, fptr, fptr]),
    "bn_merged_composite_forward": (C.c_int, [fptr, fptr, fptr, fptr, C.c_int32, C.c_int32, C.c_int32, C.c_int64, fptr, fptr, fptr,
                                              fptr, fptr, fptr, fptr, fptr, fptr, fptr, fptr]),
    "bn_merged_composite_backward": (C.c_int, [fptr, fptr, fptr, fptr, C.c_int32, C.c_int32, C.c_int32, C.c_int64, fptr, fptr, fptr,
                                               fptr, C.c_float, fptr, fptr, fptr, fptr, fptr, fptr, fptr]),
    "bn_ray_shade_loss": (C.c_int, [fptr, fptr, fptr, fptr, fptr, fptr, C.c_int64, fptr, C.c_int64, fptr, fptr, C.c_int64, fptr,
                                    C.c_int64, fptr, C.c_int64, fptr, C.c_int64, C.c_int64, fptr, fptr, fptr, C.c_int32, fptr, fptr,
                                    fptr, fptr, fptr, fptr]),
    "bn_lambert_tail": (C.c_int, [fptr, fptr, fptr, fptr, C.c_int32, C.c_int32, C.c_int32, C.c_int64, fptr, fptr, C.c_int64, fptr,
                                  C.c_int64, fptr, C.c_int64, fptr, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_int32, fptr, fptr,
                                  C.c_int32, fptr, fptr, fptr, fptr, fptr, fptr, fptr, fptr]),
    "bn_sample_brdf_forward": (C.c_int, [fptr, fptr, fptr, C.c_int64, C.c_int64, C.c_int32, C.c_int64, C.c_int64, C.c_int32, C.c_int32,
                                         fptr, C.c_int32, fptr]),
    "bn_sample_brdf_backward": (C.c_int, [fptr, fptr, fptr, C.c_int64, C.c_int64, C.c_int32, C.c_int64, C.c_int64, C.c_int32, C.c_int32,
                                          fptr, C.c_int32, fptr, fptr]),
    "bn_fold_heads": (C.c_int, [fptr, fptr]),
    "bn_unfold_heads": (C.c_int, [fptr, fptr]),
    "bn_adam_multi": (C.c_int, [fptr, fptr, fptr, fptr, C.c_int32, fptr, fptr, fptr, C.c_float, C.c_float, C.c_float, C.c_float,
                                C.c_float, C.c_int32, fptr, fptr]),
    "bn_device_faults": (C.c_int, [C.POINTER(C.c_uint), fptr]),
    "bn_prof_enable": (C.c_int, [C.c_int]),
    "bn_prof_collect": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_int]),
}

PROF_NAMES = ["pack", "field_fwd_sigma", "field_fwd_full", "field_bwd_chain", "wgrad", "skinny_wgrad", "composite_fwd",
              "composite_bwd", "guided_samples", "stratified_z", "adam", "brdf", "field_adjoint", "field_adjoint_bwd", "wgrad_reduce"]


def prof_enable(on):
    lib().bn_prof_enable(int(bool(on)))


def prof_collect():
    """-> {kernel name: (total ms, launches)} since the last collect (synchronises the recorded events)."""
    n = len(PROF_NAMES)
    ms, cnt = (C.c_double * n)(), (C.c_int * n)()
    lib().bn_prof_collect(ms, cnt, n)
    return {PROF_NAMES[i]: (ms[i], cnt[i]) for i in range(n) if cnt[i]}


def exported_symbols():
    """Every symbol include/brdfnerf_hip.h declares (checked by the CPU test-suite)."""
    return sorted(_SIGS)


def load(path, baseline=False):
    """dlopen one build of the library and bind every declared entry point (no fallback: raises if absent).
    baseline=True (measurement harnesses only, profiles/ab_kernels.py): a library built from an EARLIER commit - entry points
    it lacks stay unbound and the ABI number is not checked; the caller keeps to the calls that library has."""
    if not os.path.exists(path):
        raise LibraryMissing(f"{path} not found: build the HIP extension first "
                             f"(python -m brdf_nerf_amd.build). There is no CPU fallback.")
    L = C.CDLL(path)
    for name, (res, args) in _SIGS.items():
        if baseline and not hasattr(L, name):
            continue
        fn = getattr(L, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype, fn.argtypes = res, args
    if not baseline and L.bn_abi_version() != BN_ABI_VERSION:
        raise LibraryMissing(f"{path}: ABI version mismatch; rebuild")
    if not baseline and os.environ.get("BRDFNERF_ALLOW_STALE_LIB", "0") in ("", "0"):
        # a library older than the sources beside it must not pass for them (tests, bench and profiles all run through here)
        from .build import CSRC, source_hash
        if os.path.isdir(CSRC):
            have, want = L.bn_source_hash().decode(), source_hash()
            if have != want:
                raise LibraryMissing(f"{path} was built from other sources (library {have}, tree {want}): run "
                                     f"python -m brdf_nerf_amd.build (BRDFNERF_ALLOW_STALE_LIB=1 to load it anyway)")
    return L


def lib():
    global _lib
    if _lib is None:
        _lib = load(LIB_PATH)
        if os.environ.get("BRDFNERF_DETERMINISTIC", "0") not in ("", "0"):
            _lib.bn_set_deterministic(1)
    return _lib


def set_deterministic(on=True):
    """The reference's Trainer(deterministic=True) (main.py:726).  Since round 4 the parameter gradients are bitwise reproducible
    in EVERY mode (the weight-gradient kernels write per-split slabs that are summed in fixed order: no atomics), and since round 5
    so is the REPORTED loss of the launch-lean step (FusedTrainer.repeatable_loss: a fixed-order sum of the per-ray terms).  The
    switch is kept for callers that set it (it is part of a captured step's signature) and changes no result.  Also switched on
    by BRDFNERF_DETERMINISTIC=1.  Returns the previous setting."""
    return bool(lib().bn_set_deterministic(1 if on else 0))


def deterministic():
    """Current setting of the deterministic mode (bn_get_deterministic: a read, nothing is toggled)."""
    L = lib()
    if not hasattr(L, "bn_get_deterministic"):      # an earlier commit's library loaded by a measurement harness (load(baseline=True))
        prev = L.bn_set_deterministic(0)
        if prev:
            L.bn_set_deterministic(1)
        return bool(prev)
    return bool(L.bn_get_deterministic())


def use(handle):
    """Measurement harnesses only (profiles/ab_kernels.py): route every call through another build loaded with load(),
    so that variants are timed alternately in ONE process.  Returns the previous handle."""
    global _lib
    prev, _lib = _lib, handle
    return prev


def check(status, what=""):
    if status != 0:
        raise RuntimeError(f"{what} failed ({status}): {lib().bn_last_error().decode()}")
