"""ctypes binding of librhccq_hip.so (C ABI declared in include/rhccq.h).

The product path has NO CPU fallback: if the library is missing or a GPU is not present the
functions below raise.  (The CPU oracle lives in oracle/ and is test infrastructure only.)
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "librhccq_hip.so")

_lib = None

c_void_p, c_int32, c_int64, c_double, c_float, c_uint64 = C.c_void_p, C.c_int32, C.c_int64, C.c_double, C.c_float, C.c_uint64


class MbkProblem(C.Structure):
    """struct rhccq_mbk_problem (include/rhccq.h)."""
    _fields_ = [("off", c_int64), ("n", c_int64), ("k", c_int64), ("koff", c_int64), ("init_off", c_int64),
                ("init_n", c_int64), ("rand_off", c_int64), ("first", c_int32), ("T", c_int32)]


class ClassDesc(C.Structure):
    """struct rhccq_class_desc (include/rhccq.h): one region class of a frame for rhccq_encode_frame"""
    _fields_ = [("labels", c_void_p), ("n_seg", c_int32), ("n_region", c_int32), ("seg_region", c_void_p), ("region_bbox", c_void_p),
                ("quality", c_int32), ("reserved", c_int32)]


class FrameResult(C.Structure):
    """struct rhccq_frame_result (include/rhccq.h)"""
    _fields_ = [("n_colours", c_int32), ("index_bytes", c_int32), ("shape", c_int32 * 2), ("top_left", c_int32 * 2), ("quality3", c_int32),
                ("n_jobs", c_int32), ("ms", c_double * 8), ("class_ms", (c_double * 4) * 4)]


# name -> (restype, argtypes).  Every symbol include/rhccq.h declares is listed here; the CPU test
# suite checks that the library exports all of them.
PROTOTYPES = {
    "rhccq_abi_version": (c_int32, []),
    "rhccq_ctx_create": (c_int32, [c_int32, c_void_p, C.POINTER(c_void_p)]),
    "rhccq_ctx_destroy": (None, [c_void_p]),
    "rhccq_last_error": (C.c_char_p, [c_void_p]),
    "rhccq_ctx_set_int": (c_int32, [c_void_p, c_int32, c_int64]),
    "rhccq_sync": (c_int32, [c_void_p]),
    "rhccq_stream": (c_void_p, [c_void_p]),
    "rhccq_ctx_set_stream": (c_int32, [c_void_p, c_void_p]),
    "rhccq_params": (c_int32, [c_int64, c_double, C.POINTER(c_double), C.POINTER(c_int64)]),
    "rhccq_eps_threshold": (c_int32, [c_double, C.POINTER(c_int32), C.POINTER(c_int32), C.POINTER(c_double)]),
    "rhccq_job_scan": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, C.POINTER(c_void_p), C.POINTER(c_int32),
                                 c_int32, c_void_p, c_void_p]),
    "rhccq_job_scan_bytes": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, C.POINTER(c_void_p), C.POINTER(c_int32),
                                       c_int32, c_void_p, c_void_p]),
    "rhccq_bytemap_pack": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p]),
    "rhccq_job_set_black": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32]),
    "rhccq_bitmap_count": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p]),
    "rhccq_bitmap_emit": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rhccq_job_blackfix": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, C.POINTER(c_void_p), C.POINTER(c_int32),
                                     c_void_p, c_void_p]),
    "rhccq_job_index": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, C.POINTER(c_void_p), C.POINTER(c_int32),
                                  c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rhccq_job_index_entries": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, C.POINTER(c_void_p), C.POINTER(c_int32), c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rhccq_frame_remap_entries": (c_int32, [c_void_p, c_int32, c_int32, c_int32, C.POINTER(c_void_p), C.POINTER(c_int32), c_void_p, c_void_p, c_int32,
                                            c_void_p, c_int32]),
    "rhccq_job_stats": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, C.POINTER(c_void_p), C.POINTER(c_int32), c_void_p]),
    "rhccq_job_sort_unique_bytes": (c_int64, [c_int64]),
    "rhccq_job_sort_unique": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, C.POINTER(c_void_p), C.POINTER(c_int32), c_int32, c_void_p, c_void_p,
                                        c_int32, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rhccq_job_index_ranked": (c_int32, [c_void_p, c_int32, c_int32, c_int32, C.POINTER(c_void_p), C.POINTER(c_int32), c_void_p, c_void_p, c_void_p, c_void_p]),
    "rhccq_frame_remap_ranked": (c_int32, [c_void_p, c_int32, c_int32, c_int32, C.POINTER(c_void_p), C.POINTER(c_int32), c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_int32, c_void_p, c_int32]),
    "rhccq_eps_components": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    "rhccq_cluster_sums": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "rhccq_cluster_means": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p]),
    "rhccq_kmeans": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "rhccq_mbk_init": (c_int32, [c_void_p, c_void_p, C.POINTER(MbkProblem), c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rhccq_encode_frame": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, C.POINTER(ClassDesc), c_int32, c_void_p, c_int32, c_void_p, c_void_p,
                                     C.POINTER(FrameResult)]),
    "rhccq_npysort_head": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "rhccq_mbk_steps": (c_int32, [c_void_p, c_void_p, C.POINTER(MbkProblem), c_int32, c_int64, c_int32, c_void_p, c_int64, c_void_p,
                                  c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32]),
    "rhccq_mbk_steps_overlapped": (c_int32, [c_void_p, c_void_p, C.POINTER(MbkProblem), c_int32, c_int64, c_int32, c_void_p, c_int64, c_void_p,
                                             c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int64, C.POINTER(c_int32)]),
    "rhccq_mt_randint_host": (c_int64, [c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p]),
    "rhccq_scatter_min_host": (c_int32, [c_int64, c_void_p, c_void_p, c_int64, c_void_p]),
    "rhccq_cluster_plan_host": (c_int64, [c_void_p, c_int64, c_int64, c_int32, c_void_p, c_void_p]),
    "rhccq_merge_palettes_host": (c_int32, [c_int32, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rhccq_mbk_work_bytes": (c_int64, [C.POINTER(MbkProblem), c_int32]),
    "rhccq_px_neighbours": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, C.c_float, C.c_float, c_int32, c_void_p, c_void_p, c_void_p]),
    "rhccq_px_expand": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, C.c_float, C.c_float, c_void_p, c_void_p, c_void_p]),
    "rhccq_error_sums": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "rhccq_error_tables": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "rhccq_ssim7_blocks": (c_int64, [c_int32, c_int32]),
    "rhccq_ssim7_sums": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_int64]),
    "rhccq_split_stats_blocks": (c_int64, [c_int32, c_int32]),
    "rhccq_split_stats": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_int64, c_void_p]),
    "rhccq_slic_assign": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_double, c_int32, c_void_p]),
    "rhccq_slic_connectivity_host": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "rhccq_ccl_work_bytes": (c_int64, [c_int32, c_int32, c_int32]),
    "rhccq_ccl": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p]),
    "rhccq_ccl_select": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "rhccq_ccl_keys": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "rhccq_roi_buffer": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rhccq_edges_m2_bins": (c_int64, []),
    "rhccq_edges_gray": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "rhccq_edges_grad_hist": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    "rhccq_canny_nms": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "rhccq_edges_above": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_void_p]),
    "rhccq_label_reduce": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p]),
    "rhccq_box_count": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "rhccq_morph_dilate": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, C.POINTER(c_int32), c_int32, c_int32, c_void_p]),
    "rhccq_mask_op": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p]),
    "rhccq_gap_bridge": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int64, c_int32, c_void_p]),
    "rhccq_morph_dilate_spans": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, C.POINTER(c_int32), C.POINTER(c_int32), c_int32, c_int32,
                                           c_void_p]),
    "rhccq_box_filter_seq": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "rhccq_box_sum": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "rhccq_dist_chamfer": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    "rhccq_binary_sobel": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    "rhccq_lut_u8": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "rhccq_label_sum": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_int32, c_void_p]),
    "rhccq_masked_hist": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p]),
    "rhccq_value_mask": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p]),
    "rhccq_gauss1d_f64": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_int64, c_void_p, c_int32, c_void_p]),
    "rhccq_zoom_linear_f64": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32,
                                        c_double, c_double, c_void_p]),
    "rhccq_zoom_nearest": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    "rhccq_canny_scores_nested_bytes": (c_int64, [c_int32, c_int32]),
    "rhccq_canny_scores_nested": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_int32, c_void_p, c_int64, c_void_p]),
    "rhccq_canny_scores_bytes": (c_int64, [c_int32, c_int32, c_int32]),
    "rhccq_canny_scores": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_int64, c_void_p]),
    "rhccq_edge_score": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    "rhccq_lut_u16_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_void_p]),
    "rhccq_eps_counts": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_double, c_void_p]),
    "rhccq_eps_border": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_double, c_void_p, c_void_p]),
    "rhccq_mt_uniforms": (c_int32, [c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "rhccq_mbk_order_bytes": (c_int64, [c_int64]),
    "rhccq_mbk_order": (c_int32, [c_void_p, c_void_p, C.POINTER(MbkProblem), c_int32, c_void_p, c_void_p, c_void_p, c_int64]),
    "rhccq_mbk_assign": (c_int32, [c_void_p, c_void_p, C.POINTER(MbkProblem), c_int32, c_void_p, c_void_p, c_int64, c_void_p]),
    "rhccq_remap": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p]),
    "rhccq_frame_remap": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, C.POINTER(c_void_p), C.POINTER(c_int32),
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_int32]),
    "rhccq_merge_firstpos": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "rhccq_merge_paint": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_int32, c_void_p]),
    "rhccq_decode": (c_int32, [c_void_p, c_void_p, c_int32, c_int64, c_void_p, c_int64, c_void_p]),
    "rhccq_dct_quant": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "rhccq_luma_qstep": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_float, c_float, c_void_p, c_void_p]),
}


class RhccqError(RuntimeError):
    pass


def load():
    """Load the shared library (no GPU needed to load it) and attach prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RhccqError(f"{LIB_PATH} is missing: run `python -m roibasedimagecompression_amd.build` "
                         "(or __graft_entry__.build()); there is no CPU fallback for the product path")
    # torch bundles its own libamdhip64; it must be in the process BEFORE this library is loaded so that
    # both resolve to ONE HIP runtime (torch owns device memory and streams, this library launches on them)
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def exported_symbols():
    lib = load()
    return [n for n in PROTOTYPES if hasattr(lib, n)]
