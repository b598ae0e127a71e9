"""ctypes binding of libjubjub_hip.so (C ABI in include/jubjub_hip.h).  Fails loudly if the library is missing:
there is no CPU fallback in the product path."""
import ctypes as C
import os

# HIP maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  A context with jobs in flight uses up to six streams (launch, two
# copy streams, three MSM lanes since round 6): with four queues two of them share one and the pipelines run 1.7-2x slower (profiles/r4_pcie_inclusive.txt).  The C
# library never touches the environment (it warns once on stderr when it creates its fifth stream with fewer than 8 queues configured); this
# PYTHON package sets the variable, if the application has not, before the HIP runtime is loaded -- a HIP runtime variable, not a switch of ours.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("JJ_LIB_PATH") or os.path.join(HERE, "lib", "libjubjub_hip.so")   # JJ_LIB_PATH: A/B builds of the same library

JJ_OK, JJ_ERR_INVALID, JJ_ERR_HIP, JJ_ERR_NOMEM, JJ_ERR_NODEVICE = 0, -1, -2, -3, -4
MSM_PARTIAL_BYTES = 8256   # JJ_MSM_PARTIAL_BYTES

_vp, _sz, _u8p = C.c_void_p, C.c_size_t, C.c_void_p

# name -> argtypes after the context pointer
_SIGS = {}
for _f in ("fq", "fr"):
    for _op in ("add", "sub", "mul", "pow"):
        _SIGS["jj_%s_%s" % (_f, _op)] = [_sz, _vp, _vp, _vp]
    for _op in ("neg", "square", "double", "from_bytes_wide"):
        _SIGS["jj_%s_%s" % (_f, _op)] = [_sz, _vp, _vp]
    for _op in ("invert", "sqrt", "from_bytes"):
        _SIGS["jj_%s_%s" % (_f, _op)] = [_sz, _vp, _vp, _u8p]
for _op in ("double", "neg", "mul_by_cofactor", "to_niels"):
    _SIGS["jj_point_" + _op] = [_sz, _vp, _vp]
for _op in ("add", "sub"):
    _SIGS["jj_point_" + _op] = [_sz, _vp, _vp, _vp]
for _op in ("is_identity", "is_small_order", "is_torsion_free", "is_prime_order", "is_on_curve"):
    _SIGS["jj_" + _op] = [_sz, _vp, _u8p]
_SIGS.update({
    "jj_point_sum": [_sz, _vp, _vp],
    "jj_varbase_mul": [_sz, _vp, _vp, _vp],
    "jj_varbase_mul_exact": [_sz, _vp, _vp, _vp],
    "jj_varbase_mul_ct": [_sz, _vp, _vp, _vp],
    "jj_varbase_mul_vartime": [_sz, _vp, _vp, _vp],
    "jj_varbase_mul_vartime_compressed": [_sz, _vp, _vp, _vp],
    "jj_varbase_mul_scalar": [_sz, _vp, _vp, _vp],
    "jj_varbase_mul_compressed": [_sz, _vp, _vp, _vp],
    "jj_fixedbase_mul_compressed": [_vp, _sz, _vp, _vp],
    "jj_fixedbase_table_destroy": [_vp],
    "jj_fixedbase_mul": [_vp, _sz, _vp, _vp],
    "jj_fixedbase_multi_mul": [_vp, C.c_int, _sz, _vp, _vp],
    "jj_fixedbase_composite_create": [C.c_int, _vp, C.POINTER(C.c_int), C.POINTER(_vp)],
    "jj_fixedbase_composite_mul": [_vp, _sz, _vp, _vp],
    "jj_msm": [_sz, _vp, _vp, _vp],
    "jj_msm_dev": [_sz, _vp, _vp, _vp],
    "jj_msm_begin": [_sz, _vp, _vp, C.POINTER(_vp)],
    "jj_msm_partial": [_sz, _vp, _vp, C.c_int, C.c_int, _vp],
    "jj_msm_combine_dev": [_sz, _vp, _vp],
    "jj_ctx_set_comm": [_vp, C.c_int, C.c_int, _vp],
    "jj_msm_allgather": [_sz, _vp, _vp, C.c_int, _vp],
    "jj_msm_allgather_begin": [_sz, _vp, _vp, C.c_int, C.POINTER(_vp)],
    "jj_decompress": [_sz, _vp, C.c_uint, _vp, _u8p],
    "jj_compress": [_sz, _vp, _vp],
    "jj_batch_normalize": [_sz, _vp, _vp],
    "jj_ctx_set_stream": [_vp],
    "jj_ctx_sync": [],
    "jj_ctx_use_own_stream": [],
    "jj_ctx_profile": [C.c_int],
    "jj_ctx_profile_read": [C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int)],
    "jj_peak_imad32": [C.POINTER(C.c_double)],
    "jj_peak_imad32_samples": [C.c_int, C.POINTER(C.c_double)],
    "jj_fq_to_le_bits": [_sz, _vp, _vp],
    "jj_fr_to_le_bits": [_sz, _vp, _vp],
    "jj_synth_scalars": [_sz, C.c_uint64, C.c_uint64, _vp],
    "jj_synth_bytes32": [_sz, C.c_uint64, C.c_uint64, _vp],
    "jj_random_points": [_sz, C.c_uint64, C.c_uint64, C.c_int, _vp, _vp],
    # several devices of one node: the first argument is a jj_multi*
    "jj_multi_destroy": [],
    "jj_multi_device_count": [],
    "jj_multi_varbase_mul": [_sz, _vp, _vp, _vp],
    "jj_multi_fixedbase_table_create": [_vp, C.c_int, C.POINTER(_vp)],
    "jj_multi_fixedbase_table_destroy": [_vp],
    "jj_multi_fixedbase_mul": [_vp, _sz, _vp, _vp],
    "jj_multi_decompress": [_sz, _vp, C.c_uint, _vp, _u8p],
    "jj_multi_msm": [_sz, _vp, _vp, _vp],
})

EXPORTS = sorted(list(_SIGS) + ["jj_ctx_create", "jj_ctx_destroy", "jj_last_error", "jj_version", "jj_device_info", "jj_recommended_wnaf_for_num_scalars",
                                "jj_fixedbase_table_create", "jj_fr_char_le_bits", "jj_multi_create", "jj_multi_ctx", "jj_multi_last_error",
                                "jj_msm_fold_partials", "jj_msm_finish", "jj_msm_combine",
                                "jj_host_alloc", "jj_host_free", "jj_host_register", "jj_host_unregister",
                                "jj_result_acquire", "jj_result_release", "jj_result_pool_stats",
                                "jj_plan_host_chunks", "jj_plan_msm_host_passes", "jj_ctx_set_option", "jj_ctx_get_option"])

_lib = None


def _preload_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64 (same soname as /opt/rocm's).  Two HIP runtimes in one
    process see no devices, so when torch is installed we make its copy the process-wide runtime *before* our
    library resolves libamdhip64.so.7; a later `import torch` then reuses it.  JJ_HIP_RUNTIME=system opts out."""
    import importlib.util
    import sys

    if os.environ.get("JJ_HIP_RUNTIME", "") == "system" or "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libjubjub_hip.so is not built (%s). Run `python -m jubjub_amd.build` or __graft_entry__.build(); "
            "there is no CPU fallback." % LIB_PATH)
    _preload_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    for name, args in _SIGS.items():
        fn = getattr(lib, name)
        fn.restype = C.c_int
        fn.argtypes = [_vp] + args
    lib.jj_ctx_create.restype = C.c_int
    lib.jj_ctx_create.argtypes = [C.c_int, C.POINTER(_vp)]
    lib.jj_ctx_destroy.restype = C.c_int
    lib.jj_ctx_destroy.argtypes = [_vp]
    lib.jj_last_error.restype = C.c_char_p
    lib.jj_last_error.argtypes = [_vp]
    lib.jj_version.restype = C.c_int
    lib.jj_version.argtypes = []
    lib.jj_recommended_wnaf_for_num_scalars.restype = C.c_int
    lib.jj_recommended_wnaf_for_num_scalars.argtypes = [C.c_size_t]
    lib.jj_multi_create.restype = C.c_int
    lib.jj_multi_create.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(_vp)]
    lib.jj_multi_ctx.restype = _vp
    lib.jj_multi_ctx.argtypes = [_vp, C.c_int]
    lib.jj_multi_last_error.restype = C.c_char_p
    lib.jj_multi_last_error.argtypes = [_vp]
    lib.jj_msm_fold_partials.restype = C.c_int
    lib.jj_msm_fold_partials.argtypes = [C.c_size_t, _vp, _vp]
    lib.jj_msm_finish.restype = C.c_int
    lib.jj_msm_finish.argtypes = [_vp, _vp]
    lib.jj_msm_combine.restype = C.c_int
    lib.jj_msm_combine.argtypes = [C.c_size_t, _vp, _vp]
    lib.jj_host_alloc.restype = C.c_int
    lib.jj_host_alloc.argtypes = [C.c_size_t, C.POINTER(_vp)]
    lib.jj_result_acquire.restype = C.c_int
    lib.jj_result_acquire.argtypes = [_vp, C.c_size_t, C.POINTER(_vp)]
    lib.jj_result_release.restype = C.c_int
    lib.jj_result_release.argtypes = [_vp, _vp]
    lib.jj_result_pool_stats.restype = C.c_int
    lib.jj_result_pool_stats.argtypes = [_vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    lib.jj_host_free.restype = C.c_int
    lib.jj_host_free.argtypes = [_vp]
    lib.jj_host_register.restype = C.c_int
    lib.jj_host_register.argtypes = [_vp, C.c_size_t]
    lib.jj_host_unregister.restype = C.c_int
    lib.jj_host_unregister.argtypes = [_vp]
    lib.jj_plan_host_chunks.restype = C.c_int
    lib.jj_plan_host_chunks.argtypes = [C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.POINTER(C.c_size_t), C.c_size_t, C.POINTER(C.c_size_t)]
    lib.jj_plan_msm_host_passes.restype = C.c_int
    lib.jj_plan_msm_host_passes.argtypes = [C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    lib.jj_ctx_set_option.restype = C.c_int
    lib.jj_ctx_set_option.argtypes = [_vp, C.c_char_p, C.c_longlong]
    lib.jj_ctx_get_option.restype = C.c_int
    lib.jj_ctx_get_option.argtypes = [_vp, C.c_char_p, C.POINTER(C.c_longlong)]
    lib.jj_fr_char_le_bits.restype = C.c_int
    lib.jj_fr_char_le_bits.argtypes = [C.POINTER(C.c_uint8)]
    lib.jj_device_info.restype = C.c_int
    lib.jj_device_info.argtypes = [_vp, C.POINTER(C.c_int64)]
    lib.jj_fixedbase_table_create.restype = C.c_int
    lib.jj_fixedbase_table_create.argtypes = [_vp, _vp, C.c_int, C.POINTER(_vp)]
    _lib = lib
    return lib
