"""ctypes mirror of include/rpt.h (the C ABI).  Plain data only."""
import ctypes as C

RPT_ABI_VERSION = 4

RPT_OK = 0
RPT_ERR_INVALID_ARG = -1
RPT_ERR_NO_DEVICE = -2
RPT_ERR_HIP = -3
RPT_ERR_NO_SCENE = -4
RPT_ERR_UNSUPPORTED = -5
RPT_ERR_RCCL = -6

RPT_MAT_RGB = 1 << 0
RPT_MAT_EMISSION = 1 << 1
RPT_MAT_ANISOTROPIC = 1 << 2
RPT_MAT_METALLIC = 1 << 3
RPT_MAT_ROUGHNESS = 1 << 4
RPT_MAT_SUBSURFACE = 1 << 5
RPT_MAT_SPECULAR_TINT = 1 << 6
RPT_MAT_SHEEN = 1 << 7
RPT_MAT_SHEEN_TINT = 1 << 8
RPT_MAT_CLEARCOAT = 1 << 9
RPT_MAT_CLEARCOAT_GLOSS = 1 << 10
RPT_MAT_SPEC_TRANS = 1 << 11
RPT_MAT_IOR = 1 << 12
RPT_MAT_ALL = (1 << 13) - 1
RPT_MAT_MEDIUM = 1 << 13

RPT_MEDIUM_NONE, RPT_MEDIUM_ABSORB, RPT_MEDIUM_SCATTER, RPT_MEDIUM_EMISSIVE = range(4)

RPT_PROC_NONE = 0
RPT_PROC_CHECKER_DIR = 1

RPT_LIGHT_RECTANGULAR = 0
RPT_LIGHT_SPHERICAL = 1
RPT_LIGHT_DISTANT = 2

RPT_BG_CONSTANT = 0
RPT_BG_GRADIENT_Y = 1

RPT_SCENE_ANYHIT_USES_MAX_DIST = 1 << 0
RPT_SCENE_SAMPLE_ALL_LIGHT_TYPES = 1 << 1
RPT_SCENE_MEDIA = 1 << 2

RPT_RENDER_DEFAULT = 0
RPT_RENDER_NESTED_LOOPS = 1 << 0
RPT_RENDER_FAST_MATH = 1 << 1
RPT_RENDER_RUSSIAN_ROULETTE = 1 << 5
RPT_RENDER_SMALL_COMPACT = 1 << 8
RPT_RENDER_ALL_FLAGS = RPT_RENDER_NESTED_LOOPS | RPT_RENDER_FAST_MATH | RPT_RENDER_RUSSIAN_ROULETTE | RPT_RENDER_SMALL_COMPACT   # (the other bits: reserved)

(RPT_PROBE_SIN, RPT_PROBE_COS, RPT_PROBE_LOG2, RPT_PROBE_POW, RPT_PROBE_DIV, RPT_PROBE_SQRT, RPT_PROBE_RNG, RPT_PROBE_EXP,
 RPT_PROBE_LOG, RPT_PROBE_DIV3) = range(10)
(RPT_PROBE_FN_GEN_RAY, RPT_PROBE_FN_HIT_SPHERE, RPT_PROBE_FN_HIT_PLANE, RPT_PROBE_FN_SAMPLE_LIGHT, RPT_PROBE_FN_DISNEY_EVAL,
 RPT_PROBE_FN_DISNEY_SAMPLE, RPT_PROBE_FN_COUNT) = range(7)
RPT_PROBE_IN_STRIDE = 32
RPT_PROBE_OUT_STRIDE = 16
RPT_UNIQUE_ID_BYTES = 128

F3 = C.c_float * 3
F4 = C.c_float * 4


class rpt_material(C.Structure):
    _fields_ = [
        ("mask", C.c_uint32), ("proc_kind", C.c_uint32),
        ("rgb", F3), ("emission", F3),
        ("anisotropic", C.c_float), ("metallic", C.c_float), ("roughness", C.c_float),
        ("subsurface", C.c_float), ("specular_tint", C.c_float), ("sheen", C.c_float),
        ("sheen_tint", C.c_float), ("clearcoat", C.c_float), ("clearcoat_gloss", C.c_float),
        ("spec_trans", C.c_float), ("ior", C.c_float),
        ("proc_params", F4),
        ("medium_type", C.c_uint32), ("medium_density", C.c_float), ("medium_color", F3), ("medium_anisotropy", C.c_float),
    ]


class rpt_sphere(C.Structure):
    _fields_ = [("center", F3), ("radius", C.c_float), ("material", C.c_uint32)]


class rpt_plane(C.Structure):
    _fields_ = [("normal", F3), ("point", F3), ("min_denom", C.c_float), ("material", C.c_uint32), ("max_t", C.c_float)]


class rpt_light(C.Structure):
    _fields_ = [("type", C.c_uint32), ("position", F3), ("emission", F3), ("u", F3), ("v", F3),
                ("radius", C.c_float), ("area", C.c_float)]


class rpt_camera(C.Structure):
    _fields_ = [("origin", F3), ("center", F3), ("fov_deg", C.c_float)]


class rpt_background(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("colour_a", F3), ("colour_b", F3), ("gamma", C.c_float), ("scale", C.c_float)]


RPT_SDF_SPHERE = 0
RPT_SDF_TORUS_Y = 1


class rpt_sdf_prim(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("center", F3), ("params", C.c_float * 2)]


class rpt_sdf(C.Structure):
    _fields_ = [("n_prims", C.c_uint32), ("max_steps", C.c_uint32), ("material", C.c_uint32), ("smooth_k", C.c_float),
                ("hit_eps", C.c_float), ("max_t", C.c_float), ("normal_eps", C.c_float), ("prims", C.POINTER(rpt_sdf_prim))]


class rpt_scene_desc(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32), ("flags", C.c_uint32),
        ("camera", rpt_camera), ("background", rpt_background),
        ("eps", C.c_float), ("max_depth", C.c_uint32),
        ("n_spheres", C.c_uint32), ("spheres", C.POINTER(rpt_sphere)),
        ("n_planes", C.c_uint32), ("planes", C.POINTER(rpt_plane)),
        ("n_lights", C.c_uint32), ("lights", C.POINTER(rpt_light)),
        ("n_materials", C.c_uint32), ("materials", C.POINTER(rpt_material)),
        ("sdf", rpt_sdf),
    ]


class rpt_tile_plan(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("full_blocks", "block_rows", "host_row0", "host_row_stride", "ragged_rows",
                                           "ragged_host_row0", "ragged_tile_row0")]


class rpt_unique_id(C.Structure):
    _fields_ = [("bytes", C.c_char * RPT_UNIQUE_ID_BYTES)]


# every symbol include/rpt.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "rpt_sizeof_scene_desc": (C.c_uint32, []),
    "rpt_build_has_test_hooks": (C.c_uint32, []),
    "rpt_create_multi": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int]),
    "rpt_comm_unique_id": (C.c_int, [C.POINTER(rpt_unique_id)]),
    "rpt_create_rank": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.POINTER(rpt_unique_id)]),
    "rpt_world": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "rpt_set_tile_rows": (C.c_int, [C.c_void_p, C.c_uint32]),
    "rpt_set_dispatch": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "rpt_resident_gather_device": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rpt_resident_sync": (C.c_int, [C.c_void_p]),
    "rpt_resident_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64]),
    "rpt_resident_kernel_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "rpt_tile_rows_padded": (C.c_uint32, [C.c_uint32, C.c_uint32, C.c_uint32]),
    "rpt_tile_copy_plan": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(rpt_tile_plan)]),
    "rpt_scene_analytical": (C.c_int, [C.POINTER(rpt_scene_desc)]),
    "rpt_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "rpt_destroy": (None, [C.c_void_p]),
    "rpt_last_error": (C.c_char_p, [C.c_void_p]),
    "rpt_abi_version": (C.c_uint32, []),
    "rpt_upload_scene": (C.c_int, [C.c_void_p, C.POINTER(rpt_scene_desc)]),
    "rpt_render": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32]),
    "rpt_resident_render": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32]),
    "rpt_resident_frames": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "rpt_resident_download": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rpt_resident_download_u8": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rpt_resident_reset": (C.c_int, [C.c_void_p]),
    "rpt_host_pin": (C.c_int, [C.c_void_p, C.c_size_t]),
    "rpt_host_unpin": (C.c_int, [C.c_void_p]),
    "rpt_render_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint64,
                                    C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]),
    "rpt_tile_row_count": (C.c_uint32, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "rpt_tile_global_row": (C.c_uint32, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "rpt_untile_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                    C.c_uint32, C.c_void_p]),
    "rpt_convert_to_u8_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]),
    "rpt_convert_to_u8_at_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32,
                                              C.c_uint32, C.c_uint32, C.c_void_p]),
    "rpt_convert_to_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]),
    "rpt_denoise_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_void_p]),
    "rpt_denoise": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float]),
    "rpt_synchronize": (C.c_int, [C.c_void_p, C.c_void_p]),
}

# include/rpt_test.h: the test hooks, exported by librpt_hip_test.so only
TEST_SYMBOLS = {
    "rpt_debug_reload_knobs": (C.c_int, []),
    "rpt_probe_fn": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    "rpt_probe_rays": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p]),
    "rpt_debug_render_overlap_ms": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "rpt_debug_sched_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(C.c_uint32)]),
    "rpt_debug_kernel_choice": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "rpt_probe_math": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
}
