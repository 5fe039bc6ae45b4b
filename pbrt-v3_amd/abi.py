"""ctypes mirror of include/pbrt_gpu.h and include/pbrt_host.h.

Layouts must match the C headers field for field; tests/test_abi.py checks
sizeof() of every struct against values compiled from the headers.
"""
import ctypes as C

PG_ABI_VERSION = 29
PG_SAMPLER_HALTON, PG_SAMPLER_SOBOL, PG_SAMPLER_RANDOM, PG_SAMPLER_STRATIFIED, PG_SAMPLER_ZEROTWO, PG_SAMPLER_MAXMINDIST = range(6)
PG_OK = 0
PG_MEM_HOST, PG_MEM_DEVICE = 0, 1
PG_OPT_OVERLAP_SHADOW = 1
PG_SHADING_MATERIAL_PREPASS, PG_SHADING_LISTS_DID_NOT_FIT = 0x100, 0x200
PG_LIGHTS_UNIFORM, PG_LIGHTS_POWER, PG_LIGHTS_SPATIAL = 0, 1, 2
PG_MAT_NONE, PG_MAT_MATTE, PG_MAT_PLASTIC, PG_MAT_MIRROR, PG_MAT_GLASS = 0, 1, 2, 3, 4
PG_LIGHT_AREA, PG_LIGHT_POINT, PG_LIGHT_SPOT, PG_LIGHT_DISTANT, PG_LIGHT_INFINITE = 0, 1, 2, 3, 4
PG_TRI_FLIP_NORMAL, PG_TRI_REVERSE_ORIENTATION, PG_TRI_HAS_N, PG_TRI_HAS_UV, PG_TRI_HAS_S = 1, 2, 4, 8, 16
PG_PRIM_SPHERE, PG_PRIM_INSTANCE, PG_TRI_ALPHA = 32, 64, 128  # (PG_PRIM_INSTANCE: also inside an object definition since ABI 29)


class PgBVHNode(C.Structure):
    _fields_ = [("bmin", C.c_float * 3), ("bmax", C.c_float * 3), ("offset", C.c_int32),
                ("nprims", C.c_uint16), ("axis", C.c_uint8), ("pad", C.c_uint8)]


class PgMaterial(C.Structure):
    _fields_ = [("type", C.c_int32), ("kd", C.c_float * 3), ("ks", C.c_float * 3), ("sigma", C.c_float),
                ("roughness", C.c_float), ("remap_roughness", C.c_int32), ("kr", C.c_float * 3), ("kt", C.c_float * 3),
                ("eta", C.c_float), ("first_bxdf", C.c_int32), ("n_bxdfs", C.c_int32), ("bsdf_eta", C.c_float), ("textured_index", C.c_int32)]


class PgLight(C.Structure):
    _fields_ = [("type", C.c_int32), ("prim", C.c_int32), ("L", C.c_float * 3), ("two_sided", C.c_int32), ("area", C.c_float),
                ("pos", C.c_float * 3), ("w2l", C.c_float * 9), ("cos_total_width", C.c_float), ("cos_falloff_start", C.c_float),
                ("world_radius", C.c_float), ("l2w", C.c_float * 9), ("env_image", C.c_int32), ("env_nu", C.c_int32), ("env_nv", C.c_int32),
                ("env_table", C.c_int64), ("env_power", C.c_float * 3), ("proj", C.c_float * 16), ("screen", C.c_float * 4), ("hither", C.c_float)]


class PgTexRef(C.Structure):
    _fields_ = [("tex", C.c_int32), ("v", C.c_float * 3)]


class PgTexture(C.Structure):
    _fields_ = [("type", C.c_int32), ("is_float", C.c_int32), ("mapping", C.c_int32), ("su", C.c_float), ("sv", C.c_float),
                ("du", C.c_float), ("dv", C.c_float), ("vs", C.c_float * 3), ("vt", C.c_float * 3), ("w2t", C.c_float * 16),
                ("tex1", PgTexRef), ("tex2", PgTexRef), ("amount", PgTexRef), ("aa_none", C.c_int32),
                ("v00", C.c_float * 3), ("v01", C.c_float * 3), ("v10", C.c_float * 3), ("v11", C.c_float * 3), ("image", C.c_int32),
                ("octaves", C.c_int32), ("omega", C.c_float), ("noise_scale", C.c_float), ("variation", C.c_float)]


class PgImage(C.Structure):
    _fields_ = [("is_float", C.c_int32), ("n_levels", C.c_int32), ("width", C.c_int32), ("height", C.c_int32), ("wrap", C.c_int32),
                ("trilinear", C.c_int32), ("max_anisotropy", C.c_float), ("level_offset", C.c_int64 * 32)]


class PgMedium(C.Structure):
    _fields_ = [("sigma_a", C.c_float * 3), ("sigma_s", C.c_float * 3), ("sigma_t", C.c_float * 3), ("g", C.c_float)]


class PgDensityGrid(C.Structure):
    _fields_ = [("nx", C.c_int32), ("ny", C.c_int32), ("nz", C.c_int32), ("reserved", C.c_int32), ("density_offset", C.c_int64),
                ("sigma_t", C.c_float), ("inv_max_density", C.c_float), ("world_to_medium", C.c_float * 16)]


class PgBSSRDF(C.Structure):
    _fields_ = [("eta", C.c_float), ("sigma_t", C.c_float * 3), ("rho", C.c_float * 3), ("n_rho", C.c_int32), ("n_radius", C.c_int32),
                ("table", C.c_int64), ("textured", C.c_int32), ("scale", C.c_float), ("a", PgTexRef), ("b", PgTexRef), ("match_material", C.c_int32)]


class PgAlphaMask(C.Structure):
    _fields_ = [("has_alpha", C.c_int32), ("has_shadow_alpha", C.c_int32), ("alpha", PgTexRef), ("shadow_alpha", PgTexRef)]


class PgTexturedMaterial(C.Structure):
    _fields_ = [("kind", C.c_int32), ("s", PgTexRef * 5), ("f", PgTexRef * 4), ("has_u", C.c_int32), ("has_v", C.c_int32),
                ("remap_roughness", C.c_int32), ("sub", C.c_int32 * 2), ("has_bump", C.c_int32), ("bump", PgTexRef)]


class PgBxDF(C.Structure):
    _fields_ = [("type", C.c_int32), ("fresnel", C.c_int32), ("R", C.c_float * 3), ("T", C.c_float * 3),
                ("eta_a", C.c_float), ("eta_b", C.c_float), ("cond_eta", C.c_float * 3), ("cond_k", C.c_float * 3),
                ("alpha_x", C.c_float), ("alpha_y", C.c_float), ("on_a", C.c_float), ("on_b", C.c_float),
                ("n_scales", C.c_int32), ("scale", C.c_float * 9)]


class PgSphere(C.Structure):
    _fields_ = [("o2w", C.c_float * 16), ("w2o", C.c_float * 16), ("radius", C.c_float), ("z_min", C.c_float), ("z_max", C.c_float),
                ("theta_min", C.c_float), ("theta_max", C.c_float), ("phi_max", C.c_float),
                ("reverse_orientation", C.c_int32), ("swaps_handedness", C.c_int32),
                ("shape", C.c_int32), ("height", C.c_float), ("inner_radius", C.c_float), ("area", C.c_float),
                ("p1", C.c_float * 3), ("p2", C.c_float * 3), ("ah", C.c_float), ("ch", C.c_float)]


class PgObject(C.Structure):
    _fields_ = [("first_node", C.c_int32), ("n_nodes", C.c_int32), ("first_prim", C.c_int32), ("n_prims", C.c_int32)]


class PgInstance(C.Structure):
    _fields_ = [("i2w", C.c_float * 16), ("w2i", C.c_float * 16), ("object", C.c_int32), ("identity", C.c_int32),
                ("animated", C.c_int32), ("time", C.c_float * 2), ("i2w_end", C.c_float * 16), ("w2i_end", C.c_float * 16),
                ("T", (C.c_float * 3) * 2), ("R", (C.c_float * 4) * 2), ("S", (C.c_float * 9) * 2)]


class PgSceneDesc(C.Structure):
    _fields_ = [("abi_version", C.c_int32),
                ("n_nodes", C.c_int32), ("nodes", C.POINTER(PgBVHNode)),
                ("n_tris", C.c_int32), ("indices", C.POINTER(C.c_int32)), ("tri_flags", C.POINTER(C.c_uint32)),
                ("tri_material", C.POINTER(C.c_int32)), ("tri_light", C.POINTER(C.c_int32)),
                ("n_verts", C.c_int32), ("P", C.POINTER(C.c_float)), ("N", C.POINTER(C.c_float)),
                ("UV", C.POINTER(C.c_float)), ("S", C.POINTER(C.c_float)),
                ("n_materials", C.c_int32), ("materials", C.POINTER(PgMaterial)),
                ("n_lights", C.c_int32), ("lights", C.POINTER(PgLight)),
                ("light_strategy", C.c_int32),
                ("n_perm_dims", C.c_int32), ("perms", C.POINTER(C.c_uint16)), ("perm_sums", C.POINTER(C.c_int32)),
                ("n_spheres", C.c_int32), ("spheres", C.POINTER(PgSphere)),
                ("n_bxdfs", C.c_int32), ("bxdfs", C.POINTER(PgBxDF)),
                ("n_nodes_all", C.c_int32), ("n_prims_all", C.c_int32), ("n_objects", C.c_int32), ("objects", C.POINTER(PgObject)),
                ("n_instances", C.c_int32), ("instances", C.POINTER(PgInstance)),
                ("n_textures", C.c_int32), ("textures", C.POINTER(PgTexture)), ("n_textured", C.c_int32), ("textured", C.POINTER(PgTexturedMaterial)),
                ("n_images", C.c_int32), ("images", C.POINTER(PgImage)), ("n_texel_floats", C.c_int64), ("texels", C.POINTER(C.c_float)),
                ("n_media", C.c_int32), ("media", C.POINTER(PgMedium)), ("tri_medium_inside", C.POINTER(C.c_int32)), ("tri_medium_outside", C.POINTER(C.c_int32)),
                ("n_alphas", C.c_int32), ("alphas", C.POINTER(PgAlphaMask)), ("tri_alpha", C.POINTER(C.c_int32)),
                ("n_env_floats", C.c_int64), ("env_tables", C.POINTER(C.c_float)), ("ewa_lut", C.POINTER(C.c_float)),
                ("sobol_matrices", C.POINTER(C.c_uint32)), ("vdc_sobol", C.POINTER(C.c_uint64)), ("vdc_sobol_inv", C.POINTER(C.c_uint64)),
                ("noise_perm", C.POINTER(C.c_int32)), ("cmaxmin", C.POINTER(C.c_uint32)),
                ("n_grids", C.c_int32), ("grids", C.POINTER(PgDensityGrid)), ("media_grid", C.POINTER(C.c_int32)),
                ("n_density_floats", C.c_int64), ("grid_density", C.POINTER(C.c_float)),
                ("n_bssrdfs", C.c_int32), ("bssrdfs", C.POINTER(PgBSSRDF)), ("material_bssrdf", C.POINTER(C.c_int32)),
                ("n_bssrdf_floats", C.c_int64), ("bssrdf_tables", C.POINTER(C.c_float))]


class PgRenderDesc(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("integrator", C.c_int32), ("camera_medium", C.c_int32), ("camera_type", C.c_int32),
                ("raster_to_camera", C.c_float * 16), ("dx_camera", C.c_float * 3), ("dy_camera", C.c_float * 3), ("camera_to_world", C.c_float * 16),
                ("lens_radius", C.c_float), ("focal_distance", C.c_float),
                ("shutter_open", C.c_float), ("shutter_close", C.c_float),
                ("camera_animated", C.c_int32), ("camera_time", C.c_float * 2), ("camera_to_world_end", C.c_float * 16),
                ("camera_T", (C.c_float * 3) * 2), ("camera_R", (C.c_float * 4) * 2), ("camera_S", (C.c_float * 9) * 2),
                ("full_res", C.c_int32 * 2), ("cropped_pixel_bounds", C.c_int32 * 4), ("sample_bounds", C.c_int32 * 4),
                ("filter_radius", C.c_float * 2), ("filter_general", C.c_int32), ("tile_halo", C.c_int32 * 4),
                ("tile_pixels", C.c_int32), ("filter_table", C.c_float * 256), ("film_scale", C.c_float), ("max_sample_luminance", C.c_float),
                ("spp", C.c_int32), ("base_scales", C.c_int32 * 2), ("base_exponents", C.c_int32 * 2),
                ("sample_stride", C.c_int32), ("mult_inverse", C.c_int32 * 2), ("sample_at_pixel_center", C.c_int32),
                ("sampler", C.c_int32), ("sobol_resolution", C.c_int32), ("sobol_log2_resolution", C.c_int32),
                ("sampler_dims", C.c_int32), ("strat_samples", C.c_int32 * 2), ("strat_jitter", C.c_int32),
                ("max_depth", C.c_int32), ("rr_threshold", C.c_float), ("pixel_bounds", C.c_int32 * 4),
                ("tile_first", C.c_int32), ("tile_step", C.c_int32)]


class PgFilmPixel(C.Structure):
    _fields_ = [("rgb", C.c_float * 3), ("weight", C.c_float)]


class PgStraySample(C.Structure):
    _fields_ = [("px", C.c_int32), ("py", C.c_int32), ("src_px", C.c_int32), ("src_py", C.c_int32),
                ("rgb", C.c_float * 3), ("weight", C.c_float)]


class PgCounters(C.Structure):
    _fields_ = [("camera_rays", C.c_uint64), ("closest_rays", C.c_uint64), ("shadow_rays", C.c_uint64),
                ("node_visits", C.c_uint64), ("tri_tests", C.c_uint64), ("light_tri_tests", C.c_uint64),
                ("closest_node_visits", C.c_uint64), ("closest_tri_tests", C.c_uint64),
                ("shadow_node_visits", C.c_uint64), ("shadow_tri_tests", C.c_uint64),
                ("closest_launches", C.c_uint64), ("shadow_launches", C.c_uint64),
                ("closest_ms", C.c_double), ("shadow_ms", C.c_double), ("render_ms", C.c_double),
                ("shade_launches", C.c_uint64), ("resolve_launches", C.c_uint64), ("shade_items", C.c_uint64),
                ("mis_rays", C.c_uint64), ("shade_ms", C.c_double), ("resolve_ms", C.c_double),
                ("generate_ms", C.c_double), ("film_ms", C.c_double), ("shading_modes", C.c_uint64),
                ("paths_total", C.c_uint64), ("paths_zero_radiance", C.c_uint64), ("path_length_sum", C.c_uint64), ("path_length_count", C.c_uint64),
                ("path_length_min", C.c_uint64), ("path_length_max", C.c_uint64), ("volume_interactions", C.c_uint64), ("surface_interactions", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


# every symbol include/pbrt_gpu.h declares: name -> (restype, argtypes)
GPU_SYMBOLS = {
    "pg_device_count": (C.c_int, []),
    "pg_set_device": (C.c_int, [C.c_int]),
    "pg_last_error": (C.c_char_p, []),
    "pg_scene_create": (C.c_int, [C.POINTER(PgSceneDesc), C.POINTER(C.c_void_p)]),
    "pg_scene_destroy": (None, [C.c_void_p]),
    "pg_render_tile_count": (C.c_int, [C.POINTER(PgRenderDesc)]),
    "pg_render": (C.c_int, [C.c_void_p, C.POINTER(PgRenderDesc), C.c_void_p, C.c_void_p, C.c_int32,
                            C.c_void_p, C.c_int, C.c_void_p]),
    "pg_render_sharded": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32, C.POINTER(PgRenderDesc), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                    C.c_int32, C.POINTER(C.c_int32)]),
    "pg_shard_transport": (C.c_char_p, []),
    "pg_box_filter_needs_gather": (C.c_int, [C.POINTER(PgRenderDesc)]),
    "pg_intersect": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                               C.c_void_p, C.c_int, C.c_void_p]),
    "pg_intersect_p": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                 C.c_void_p]),
    "pg_counters": (C.c_int, [C.c_void_p, C.POINTER(PgCounters)]),
    "pg_counters_reset": (C.c_int, [C.c_void_p]),
    "pg_scene_set_option": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "pg_hlbvh_build": (C.c_int, [C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(C.c_int32), C.c_void_p]),
}

HOST_SYMBOLS = {
    "pbrt_host_load_file": (C.c_void_p, [C.c_char_p, C.c_int, C.POINTER(C.c_float)]),
    "pbrt_host_load_string": (C.c_void_p, [C.c_char_p, C.c_int, C.POINTER(C.c_float)]),
    "pbrt_host_free": (None, [C.c_void_p]),
    "pbrt_host_scene_desc": (C.POINTER(PgSceneDesc), [C.c_void_p]),
    "pbrt_host_render_desc": (None, [C.c_void_p, C.POINTER(PgRenderDesc)]),
    "pbrt_host_film_size": (None, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "pbrt_host_film_clear": (None, [C.c_void_p]),
    "pbrt_host_film_merge": (None, [C.c_void_p, C.POINTER(PgRenderDesc), C.c_void_p, C.c_void_p, C.c_int]),
    "pbrt_host_film_merge_shards": (None, [C.c_void_p, C.POINTER(PgRenderDesc), C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int)]),
    "pbrt_host_film_image": (None, [C.c_void_p, C.c_void_p]),
    "pbrt_host_write_pfm": (C.c_int, [C.c_char_p, C.c_void_p, C.c_int, C.c_int]),
    "pbrt_host_write_image": (C.c_int, [C.c_char_p, C.c_void_p, C.c_int, C.c_int]),
    "pbrt_host_write_image_window": (C.c_int, [C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "pbrt_host_error_count": (C.c_int, []),
    "pbrt_host_motion_bounds": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    "pbrt_host_hlbvh_build": (None, [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int), C.c_void_p]),
    "pbrt_host_set_device_bvh": (None, [C.c_int]),
}


def bind(lib, table):
    """Attach restype/argtypes for every symbol; raises AttributeError if one is missing."""
    for name, (res, args) in table.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib
