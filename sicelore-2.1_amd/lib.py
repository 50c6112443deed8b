"""ctypes binding of libsicelore_mi.so (C ABI: include/sicelore_mi.h).

PyTorch is used by callers only for device memory and streams; tensors cross this boundary as raw
``data_ptr()`` integers.  Nothing here computes: a missing library is a hard error.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

# mirrors of the C structs (include/sicelore_mi.h)
BC_WINDOW_DTYPE = np.dtype([("bases", "<u8"), ("nmask", "<u4"), ("flags", "<u4")])
BC_RESULT_DTYPE = np.dtype(
    [("bc", "<u4"), ("ed_sec", "<i4"), ("found", "i1"), ("ed", "i1"), ("offset", "i1"), ("ins_minus_del", "i1"),
     ("n_matches", "<u4")]
)
SCAN_RESULT_DTYPE = np.dtype(
    [("flags", "<u4"), ("adapter_start", "<i4"), ("adapter_end", "<i4"), ("polya_start", "<i4"), ("polya_end", "<i4"),
     ("scan_end", "<i2"), ("tso_start", "<i2"), ("adapter_nmis", "<i2"), ("found", "i1"), ("reverse", "i1"),
     ("pass1_ok", "i1"), ("reserved", "i1"), ("tso_end", "<i2")]
)
SCAN_CONFIG_DTYPE = np.dtype(
    [("min_read_length", "<i4"), ("polya_len", "<i4"), ("polya_frac", "<f4"), ("window_polya", "<i4"),
     ("max_mismatches", "<i4"), ("min_adapter_3p_matches", "<i4"), ("min_mean_bc_qv", "<i4"),
     ("min_mean_read_qv", "<i4"), ("adapter_len", "<i4"), ("adapter4", "<u4", (22,)), ("five_prime", "<i4"),
     ("dont_search_polya", "<i4"), ("adapter_search_window", "<i4"), ("tso4", "<u4", (16,)), ("tso_window", "<i4"),
     ("tso_max_mismatches", "<i4"), ("tso_min_consec", "<i4"), ("tso_min_two_best", "<i4")]
)
CHIMERA_RESULT_DTYPE = np.dtype([("pos", "<i4", (2,)), ("n_split", "u1"), ("reason", "u1", (2,)), ("flags", "u1"),
                                 ("n_matches", "<i4")])
SPLIT_REASONS = ["REV_ADAPTER", "FWD_ADAPTER", "REV_ADAPTER_FWD_ADAPTER", "REV_ADAPTER_FWD_TSO", "REV_TSO_FWD_ADAPTER",
                 "REV_TSO_FWD_TSO"]
CHIM_MULTI, CHIM_RANGE, CHIM_OVERFLOW = 1, 2, 4
END_BASES = 224
ENDS_ROWS = 28
FLAG_BITS = {"FAILED": 5, "PASSED_FWD": 8, "PASSED_REV": 9, "POLY_T_5P": 11, "POLY_A_3P": 12, "POLY_A_NOT_FOUND": 13,
             "POLY_T_5P_POLY_A_3P": 14, "ADAPTER_5P": 15, "ADAPTER_3P": 16, "ADAPTER_SELECTED_DESP_BOTH": 19,
             "READ_TOO_SHORT": 20, "ADAPTER_5P_AND_3P": 21, "TSO_5P": 17, "TSO_3P": 18, "TSO_5P_AND_3P": 22}
assert BC_WINDOW_DTYPE.itemsize == 16 and BC_RESULT_DTYPE.itemsize == 16 and SCAN_RESULT_DTYPE.itemsize == 32
assert CHIMERA_RESULT_DTYPE.itemsize == 16

SET_USED_LIST = 0
SET_WHITELIST = 1
SET_MEMBERSHIP = 2   # the list of all possible barcodes for pass 1 only: membership pyramid, no matcher structures (smi_set_mode)

# every symbol include/sicelore_mi.h declares (tests check the export table against this list)
EXPORTS = [
    "smi_last_error", "smi_version", "smi_ctx_create", "smi_ctx_destroy", "smi_ctx_device", "smi_set_barcode_set",
    "smi_set_barcode_set_device", "smi_bc_match_batch", "smi_bc_match_device", "smi_extract_windows_device",
    "smi_hist_device", "smi_last_kernel_ms", "smi_set_timing", "smi_scan_default_config", "smi_pack_ends_device",
    "smi_ctx_set_polya", "smi_scan_device", "smi_hist_windows_device", "smi_pass1_keys_device", "smi_count_keys_device", "smi_scanfastq_pass1_chunk_keys", "smi_kernel_ms", "smi_finalize_used_list", "smi_umi_dist_device", "smi_format_read_name",
    "smi_chimera_default_config", "smi_read_planes_words", "smi_pack_reads_device", "smi_chimera_device",
    "smi_split_offsets_device", "smi_chimera_fragment_name", "smi_umi_cluster_default_config", "smi_umi_cluster_groups",
    "smi_region_group", "smi_ref_position_at_read_position", "smi_scan_default_config_5p", "smi_chimera_default_config_5p", "smi_fastq_index_device", "smi_fastq_gather_device",
    "smi_fastq_write_device", "smi_bgzf_uncompressed_size", "smi_bgzf_inflate", "smi_bam_header", "smi_bam_index_records", "smi_gz_inflate", "smi_bgzf_deflate", "smi_pass2_default_config", "smi_scanfastq_pass2_chunk", "smi_scanfastq_pass1_chunk", "smi_host_alloc", "smi_host_free", "smi_assignumis_default_config", "smi_assignumis_chunk", "smi_bc_counts_device", "smi_assigned_tsv", "smi_barcode_list_tsv", "smi_hist_allreduce", "smi_hist_allreduce_after", "smi_hist_allreduce_release", "smi_ctx_create_lane", "smi_ctx_lane_refresh", "smi_scan_batch", "smi_umi_dist_batch",
    "smi_genes_load_refflat", "smi_genes_load_gtf", "smi_genes_free", "smi_genes_count", "smi_genes_dump", "smi_gene_tag_chunk", "smi_gene_tag_bam",
    "smi_pack_reads_text_device", "smi_pack_ends_text_device", "smi_frag_text_starts_device", "smi_fastq_write_text_device",
    "smi_fastq_index_host", "smi_pack_reads_host", "smi_pack_quals_host", "smi_scanfastq_pass2_packed", "smi_fastq_write_host",
    "smi_scanfastq_pass2_chunk_packed", "smi_scanfastq_pass1_chunk_packed", "smi_ends_from_planes_device",
    "smi_packed_planes_words", "smi_fastq_index_pack_host", "smi_scanfastq_pass2_packed_seg", "smi_umi_cluster_groups_device",
    "smi_record_flags", "smi_scan_stats_add", "smi_scan_stats_merge", "smi_scan_stats_tsv", "smi_deflate_bound", "smi_gzip_device", "smi_gz_inflate_device", "smi_bgzf_deflate_device",
    "smi_gene_counts_create", "smi_gene_counts_free", "smi_gene_counts_add", "smi_gene_counts_merge", "smi_gene_counts_info",
    "smi_gene_counts_tsv", "smi_umi_depths_tsv", "smi_gz_inflate_into", "smi_gene_counts_dump", "smi_gene_counts_load", "smi_gene_counts_merge_shard",
    "smi_set_stats", "smi_umi_padded_row", "smi_umi_padded_bytes", "smi_umi_dist_device_padded", "smi_ctx_set_random_barcodes", "smi_run_knobs_default", "smi_ctx_set_knobs", "smi_ctx_get_knobs", "smi_scan_config_from_knobs", "smi_chimera_config_from_knobs",
    "smi_bam_write_default_config", "smi_bam_write_batch", "smi_bam_chunk_inputs", "smi_bam_name_seen", "smi_name_set_create", "smi_name_set_free", "smi_name_set_seen",
]


class SmiError(RuntimeError):
    pass


def library_path():
    """the shipped library; SMI_LIBRARY names a measurement / experiment build instead (csrc/Makefile: MEASURE=1, VARIANT=...),
    which the tests and the bench never set"""
    return os.environ.get("SMI_LIBRARY") or os.path.join(_HERE, "csrc", "libsicelore_mi.so")


_LIB = None


def load_library():
    """Load the HIP library; raises SmiError when it has not been built (no fallback)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # torch first: its wheel carries a HIP runtime of its own (torch/lib/libamdhip64.so); a process that loads /opt/rocm's copy before it (this library
    # is linked against libamdhip64.so.7) ends up with two runtimes and torch then finds no GPU.  Loaded in this order both share torch's.
    import torch  # noqa: F401

    path = library_path()
    if not os.path.exists(path):
        raise SmiError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950)")
    lib = ctypes.CDLL(path)
    vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
    lib.smi_last_error.restype = ctypes.c_char_p
    lib.smi_version.restype = ctypes.c_char_p
    lib.smi_ctx_create.argtypes = [ci, ctypes.POINTER(vp)]
    lib.smi_ctx_destroy.argtypes = [vp]
    lib.smi_ctx_create_lane.argtypes = [vp, ctypes.POINTER(vp)]
    lib.smi_ctx_lane_refresh.argtypes = [vp]
    lib.smi_ctx_device.argtypes = [vp]
    lib.smi_ctx_set_polya.argtypes = [vp, ci, ctypes.c_float, ci]
    lib.smi_set_stats.argtypes = [vp, vp, ci]
    lib.smi_umi_padded_row.argtypes = [ctypes.c_uint32]
    lib.smi_umi_padded_row.restype = ctypes.c_uint64
    lib.smi_umi_padded_bytes.argtypes = [ctypes.c_uint32]
    lib.smi_umi_padded_bytes.restype = ctypes.c_uint64
    lib.smi_umi_dist_device_padded.argtypes = [vp, vp, vp, vp, vp, ctypes.c_uint32, ctypes.c_uint64, vp, vp]
    lib.smi_ctx_set_random_barcodes.argtypes = [vp, ctypes.c_uint64]
    lib.smi_run_knobs_default.argtypes = [vp]
    lib.smi_ctx_set_knobs.argtypes = [vp, vp]
    lib.smi_ctx_get_knobs.argtypes = [vp, vp]
    lib.smi_scan_config_from_knobs.argtypes = [vp, ci, ci, ci, vp]
    lib.smi_chimera_config_from_knobs.argtypes = [vp, ci, vp]
    lib.smi_set_barcode_set.argtypes = [vp, vp, sz, ci]
    lib.smi_set_barcode_set_device.argtypes = [vp, vp, sz, ci, vp]
    lib.smi_bc_match_batch.argtypes = [vp, vp, sz, ci, ci, vp]
    lib.smi_bc_match_device.argtypes = [vp, vp, sz, ci, ci, vp, vp]
    lib.smi_extract_windows_device.argtypes = [vp, vp, vp, vp, sz, ci, vp, vp]
    lib.smi_hist_device.argtypes = [vp, vp, vp, sz, vp, vp]
    lib.smi_scan_default_config.argtypes = [ci, vp]
    lib.smi_scan_default_config_5p.argtypes = [ci, ci, vp]
    lib.smi_pack_ends_device.argtypes = [vp, vp, vp, vp, sz, ci, vp, vp, vp, vp, vp]
    lib.smi_scan_device.argtypes = [vp, vp, vp, vp, vp, sz, vp, vp, vp, vp]
    lib.smi_hist_windows_device.argtypes = [vp, vp, vp, sz, vp, vp]
    lib.smi_last_kernel_ms.argtypes = [vp, ctypes.POINTER(ctypes.c_float)]
    lib.smi_set_timing.argtypes = [vp, ci]
    lib.smi_kernel_ms.argtypes = [vp, ci, ctypes.POINTER(ctypes.c_float)]
    lib.smi_umi_dist_device.argtypes = [vp, vp, vp, vp, vp, ctypes.c_uint32, ctypes.c_uint64, vp, vp]
    lib.smi_chimera_default_config.argtypes = [vp]
    lib.smi_fastq_index_device.argtypes = [vp, vp, sz, vp, sz, vp, vp, vp, vp, vp, vp, sz, ctypes.POINTER(sz),
                                           ctypes.POINTER(ctypes.c_uint32), vp]
    lib.smi_fastq_gather_device.argtypes = [vp, vp, vp, vp, sz, vp, vp]
    lib.smi_bgzf_uncompressed_size.argtypes = [vp, sz, ctypes.POINTER(sz), ctypes.POINTER(sz), ctypes.POINTER(sz)]
    lib.smi_bgzf_inflate.argtypes = [vp, sz, vp, sz, ctypes.POINTER(sz), ctypes.POINTER(sz), ci]
    lib.smi_bc_counts_device.argtypes = [vp, vp, sz, vp, vp]
    lib.smi_assigned_tsv.argtypes = [vp, vp, sz, ci, vp, sz, ctypes.POINTER(sz)]
    lib.smi_assignumis_default_config.argtypes = [vp]
    lib.smi_assignumis_chunk.argtypes = [vp, vp, vp, vp, vp, vp, vp, ctypes.c_int32, vp, vp, ctypes.POINTER(ctypes.c_int32)]
    lib.smi_host_alloc.argtypes = [sz, ctypes.POINTER(vp)]
    lib.smi_host_free.argtypes = [vp]
    lib.smi_pass2_default_config.argtypes = [vp]
    lib.smi_scanfastq_pass2_chunk.argtypes = [vp, vp, sz, vp, vp]
    lib.smi_scanfastq_pass1_chunk.argtypes = [vp, vp, sz, ci, ci, vp, ctypes.POINTER(sz), ctypes.POINTER(ctypes.c_uint32)]
    lib.smi_scanfastq_pass1_chunk_keys.argtypes = [vp, vp, sz, ci, ci, vp, sz, vp, ctypes.POINTER(sz), ctypes.POINTER(ctypes.c_uint32)]
    lib.smi_pass1_keys_device.argtypes = [vp, vp, vp, sz, vp, sz, vp, vp]
    lib.smi_count_keys_device.argtypes = [vp, vp, sz, vp, vp, vp, vp]
    lib.smi_bgzf_deflate.argtypes = [vp, sz, vp, sz, ctypes.POINTER(sz), ci, ci, ci]
    lib.smi_gz_inflate.argtypes = [vp, sz, vp, sz, ctypes.POINTER(sz)]
    lib.smi_bam_header.argtypes = [vp, sz, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint32),
                                   ctypes.POINTER(ctypes.c_int32), vp, vp, vp, sz, ctypes.POINTER(ctypes.c_uint64)]
    lib.smi_bam_index_records.argtypes = [vp, sz, ctypes.c_uint64, vp, sz, ctypes.POINTER(sz), ctypes.POINTER(ctypes.c_uint64)]
    lib.smi_fastq_write_device.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, ctypes.c_uint32, vp, vp, sz, vp, sz,
                                           vp, vp, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint32), vp]
    lib.smi_fastq_write_text_device.argtypes = lib.smi_fastq_write_device.argtypes
    lib.smi_pack_reads_text_device.argtypes = [vp, vp, vp, vp, sz, ctypes.c_uint64, vp, vp]
    lib.smi_pack_ends_text_device.argtypes = [vp, vp, vp, vp, sz, vp, vp, vp]
    lib.smi_frag_text_starts_device.argtypes = [vp, vp, vp, vp, vp, vp, sz, vp, vp, vp]
    lib.smi_chimera_default_config_5p.argtypes = [vp]
    lib.smi_fastq_index_host.argtypes = [vp, sz, vp, vp, sz, ctypes.POINTER(sz), ctypes.POINTER(ctypes.c_uint32), ci]
    lib.smi_pack_reads_host.argtypes = [vp, vp, vp, sz, vp, ci]
    lib.smi_pack_quals_host.argtypes = [vp, vp, sz, ci, vp, vp, ci]
    lib.smi_scanfastq_pass2_packed.argtypes = [vp, vp, vp, sz, vp, vp]
    lib.smi_fastq_write_host.argtypes = [vp, vp, vp, vp, ctypes.c_uint32, vp, vp, sz, vp, sz, ctypes.POINTER(ctypes.c_uint64),
                                         ctypes.POINTER(ctypes.c_uint32), ci]
    lib.smi_scanfastq_pass2_chunk_packed.argtypes = [vp, vp, sz, vp, ci, vp]
    lib.smi_scanfastq_pass1_chunk_packed.argtypes = [vp, vp, sz, ci, ci, vp, ci, ctypes.POINTER(sz), ctypes.POINTER(ctypes.c_uint32)]
    lib.smi_ends_from_planes_device.argtypes = [vp, vp, vp, sz, ctypes.c_uint64, vp, vp, sz, vp, vp, vp]
    lib.smi_deflate_bound.argtypes = [sz]
    lib.smi_deflate_bound.restype = sz
    lib.smi_gzip_device.argtypes = [vp, vp, sz, vp, sz, vp, ctypes.c_int, vp]
    lib.smi_gz_inflate_device.argtypes = [vp, vp, vp, ctypes.c_int, vp, vp, vp]
    lib.smi_bgzf_deflate_device.argtypes = [vp, vp, sz, vp, sz, ctypes.POINTER(sz)]
    lib.smi_packed_planes_words.argtypes = [sz, ci]
    lib.smi_packed_planes_words.restype = sz
    lib.smi_fastq_index_pack_host.argtypes = [vp, sz, vp, vp, vp, sz, vp, sz, vp, ctypes.POINTER(sz), ctypes.POINTER(ctypes.c_uint32), ci]
    lib.smi_scanfastq_pass2_packed_seg.argtypes = [vp, vp, vp, sz, vp, vp]
    lib.smi_umi_cluster_groups_device.argtypes = [vp, vp, vp, vp, ctypes.c_uint32, vp, vp, vp, vp, vp]
    lib.smi_record_flags.argtypes = [vp, vp, ci, ci]
    lib.smi_record_flags.restype = ctypes.c_uint64
    lib.smi_scan_stats_add.argtypes = [vp, vp]
    lib.smi_scan_stats_merge.argtypes = [vp, vp]
    lib.smi_scan_stats_tsv.argtypes = [vp, vp, sz, ctypes.POINTER(sz)]
    lib.smi_read_planes_words.argtypes = [ctypes.c_uint64, sz]
    lib.smi_read_planes_words.restype = sz
    lib.smi_pack_reads_device.argtypes = [vp, vp, vp, sz, ctypes.c_uint64, vp, vp]
    lib.smi_chimera_device.argtypes = [vp, vp, vp, sz, ctypes.c_uint64, vp, vp, vp]
    lib.smi_split_offsets_device.argtypes = [vp, vp, vp, sz, vp, vp, vp, vp, vp]
    lib.smi_chimera_fragment_name.argtypes = [ctypes.c_char_p, vp, ci, ctypes.c_char_p, sz]
    lib.smi_umi_cluster_default_config.argtypes = [vp]
    lib.smi_umi_cluster_groups.argtypes = [vp, vp, vp, ctypes.c_uint32, vp, vp, vp, vp, ci]
    lib.smi_region_group.argtypes = [vp, vp, vp, ctypes.c_int32, ctypes.c_int32, ci, vp, vp]
    lib.smi_ref_position_at_read_position.argtypes = [vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, vp]
    lib.smi_format_read_name.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int32, vp, vp,
                                         ctypes.c_int32, ctypes.c_uint32, ci, ctypes.c_char_p, sz]
    lib.smi_hist_allreduce.argtypes = [vp, ci, vp, sz]
    lib.smi_hist_allreduce_after.argtypes = [vp, ci, vp, sz, vp]
    lib.smi_hist_allreduce_release.argtypes = []
    lib.smi_scan_batch.argtypes = [vp, vp, vp, vp, sz, vp, vp, vp]
    lib.smi_umi_dist_batch.argtypes = [vp, vp, vp, ctypes.c_uint32, vp]
    lib.smi_genes_load_refflat.argtypes = [vp, sz, vp, ci, vp]
    lib.smi_genes_load_gtf.argtypes = [vp, sz, vp, ci, vp]
    lib.smi_genes_free.argtypes = [vp]
    lib.smi_genes_count.argtypes = [vp, vp, vp, vp]
    lib.smi_genes_dump.argtypes = [vp, vp, sz, ctypes.POINTER(sz)]
    lib.smi_gene_tag_chunk.argtypes = [vp, vp, vp, vp, vp, vp, ctypes.c_int32, vp, sz, vp, ctypes.POINTER(sz)]
    lib.smi_gene_tag_bam.argtypes = [vp, vp, sz, vp, ctypes.c_int32, vp, sz, vp, ctypes.POINTER(sz)]
    lib.smi_barcode_list_tsv.argtypes = [vp, vp, sz, ctypes.c_uint32, ci, ci, ci, ci, vp, sz, ctypes.POINTER(sz)]
    lib.smi_gz_inflate_into.argtypes = [vp, sz, ctypes.POINTER(sz), vp, sz, ctypes.POINTER(sz)]
    lib.smi_bam_write_default_config.argtypes = [vp]
    lib.smi_bam_write_batch.argtypes = [vp, sz, vp, vp, ctypes.c_int32, vp, vp, vp, vp, vp, sz, ctypes.POINTER(sz), vp, sz, ctypes.POINTER(sz), vp,
                                        vp, vp, vp]
    lib.smi_bam_chunk_inputs.argtypes = [vp, sz, vp, vp, ctypes.c_int32, vp, vp, vp, vp, vp, vp, ctypes.POINTER(sz), ctypes.POINTER(sz)]
    lib.smi_bam_name_seen.argtypes = [vp, sz, vp, ctypes.c_int32, vp]
    lib.smi_name_set_create.argtypes = [ctypes.POINTER(vp)]
    lib.smi_name_set_free.argtypes = [vp]
    lib.smi_name_set_seen.argtypes = [vp, vp, sz, vp, ctypes.c_int32, ctypes.c_int32, vp]
    lib.smi_gene_counts_create.argtypes = [ctypes.POINTER(vp)]
    lib.smi_gene_counts_free.argtypes = [vp]
    lib.smi_gene_counts_add.argtypes = [vp, sz, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci]
    lib.smi_gene_counts_merge.argtypes = [vp, vp]
    lib.smi_gene_counts_dump.argtypes = [vp, vp, sz, ctypes.POINTER(sz)]
    lib.smi_gene_counts_load.argtypes = [vp, sz, ctypes.POINTER(vp)]
    lib.smi_gene_counts_merge_shard.argtypes = [vp, vp, ctypes.POINTER(sz)]
    lib.smi_gene_counts_info.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.smi_gene_counts_tsv.argtypes = [vp, ci, vp, sz, ctypes.POINTER(sz)]
    lib.smi_umi_depths_tsv.argtypes = [vp, vp, sz, ctypes.POINTER(sz)]
    lib.smi_finalize_used_list.argtypes = [vp, vp, sz, ctypes.c_uint32, ci, ci, ci, vp, vp, vp, ctypes.POINTER(sz)]
    explicit = {"smi_last_error", "smi_version", "smi_read_planes_words", "smi_packed_planes_words", "smi_record_flags"}  # restype set above (char*, size_t)
    for name in EXPORTS:
        if name not in explicit:
            getattr(lib, name).restype = ci
    _LIB = lib
    return lib


def _ptr(t):
    """raw address of a torch tensor / numpy array / None"""
    if t is None:
        return None
    if hasattr(t, "data_ptr"):
        return ctypes.c_void_p(t.data_ptr())
    return ctypes.c_void_p(t.ctypes.data)


def _stream_ptr(stream):
    if stream is None:
        import torch

        stream = torch.cuda.current_stream()
    return ctypes.c_void_p(stream.cuda_stream if hasattr(stream, "cuda_stream") else int(stream))


def finalize_used_list(keys, counts, record_count, merge_ed=1, min_count_fold=10, cells_fold_below_max=500):
    """End of pass 1 (host, no GPU): smi_finalize_used_list -> (keys, counts, ranks) sorted by count descending."""
    lib = load_library()
    k = np.ascontiguousarray(keys, dtype=np.uint64)
    c = np.ascontiguousarray(counts, dtype=np.uint32)
    ok, oc, orank = np.zeros(k.size, np.uint64), np.zeros(k.size, np.uint32), np.zeros(k.size, np.uint32)
    n_out = ctypes.c_size_t(0)
    rc = lib.smi_finalize_used_list(_ptr(k), _ptr(c), k.size, int(record_count), int(merge_ed), int(min_count_fold),
                                    int(cells_fold_below_max), _ptr(ok), _ptr(oc), _ptr(orank), ctypes.byref(n_out))
    if rc != 0:
        raise SmiError(f"smi_finalize_used_list error {rc}: {lib.smi_last_error().decode()}")
    m = n_out.value
    return ok[:m], oc[:m], orank[:m]


def barcode_list_tsv(keys, counts, record_count, merge_ed=1, min_count_fold=10, cells_fold_below_max=500, no_whitelist=False):
    """BarcodeList.tsv text (smi_barcode_list_tsv) from the pass-1 histogram (non-zero keys / counts, as for finalize_used_list)"""
    lib = load_library()
    k = np.ascontiguousarray(keys, dtype=np.uint64)
    c = np.ascontiguousarray(counts, dtype=np.uint32)
    n = ctypes.c_size_t(0)
    args = (_ptr(k), _ptr(c), k.size, int(record_count), int(merge_ed), int(min_count_fold), int(cells_fold_below_max), int(no_whitelist))
    if lib.smi_barcode_list_tsv(*args, None, 0, ctypes.byref(n)):
        raise SmiError(lib.smi_last_error().decode())
    out = ctypes.create_string_buffer(n.value + 1)
    if lib.smi_barcode_list_tsv(*args, out, n.value, ctypes.byref(n)):
        raise SmiError(lib.smi_last_error().decode())
    return out.raw[:n.value].decode()


class GtfText(str):
    """the text of a GTF annotation, where the `refflat=` arguments of assignumis.py / GeneTagger take the text of a refFlat file (the reference picks
    the reader by the file's name: GeneAnnotationReader.loadAnnotationsFile L46-55)"""


class GeneTagger:
    """The --annotationFile of assignumis: refFlat (or GTF: a GtfText) genes on the BAM header's reference sequences (smi_genes_load_refflat /
    smi_genes_load_gtf), and the GE / GS / XF values of records (smi_gene_tag_chunk) = GennameTagger.annotateGene
    (FJ!umifinder/bamreaders/GennameTagger.java:L73-121, L382)."""

    CIGAR_OPS = "MIDNSHP=X"

    def __init__(self, refflat_text, ref_names):
        lib = load_library()
        text = refflat_text.encode() if isinstance(refflat_text, str) else bytes(refflat_text)
        names = [n.encode() if isinstance(n, str) else n for n in ref_names]
        arr = (ctypes.c_char_p * max(len(names), 1))(*names)
        h = ctypes.c_void_p()
        load = lib.smi_genes_load_gtf if isinstance(refflat_text, GtfText) else lib.smi_genes_load_refflat
        if load(text, len(text), arr, len(names), ctypes.byref(h)):
            raise SmiError(lib.smi_last_error().decode())
        self._h, self._lib = h, lib
        a, b, c = ctypes.c_size_t(0), ctypes.c_size_t(0), ctypes.c_size_t(0)
        lib.smi_genes_count(h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
        self.n_genes, self.n_lines, self.n_skipped = a.value, b.value, c.value

    def dump(self):
        """the loaded genes (smi_genes_dump): [{name, contig, start, end, negative, transcripts: [{name, tx, cds, exons}]}] in the order the reference adds
        them to its OverlapDetector, transcripts in the order Gene.iterator() walks them"""
        n = ctypes.c_size_t(0)
        if self._lib.smi_genes_dump(self._h, None, 0, ctypes.byref(n)):
            raise SmiError(self._lib.smi_last_error().decode())
        buf = ctypes.create_string_buffer(max(n.value, 1))
        if self._lib.smi_genes_dump(self._h, buf, n.value, ctypes.byref(n)):
            raise SmiError(self._lib.smi_last_error().decode())
        out = []
        for ln in buf.raw[:n.value].decode("latin-1").split("\n"):
            if not ln:
                continue
            name, contig, st, en, strand, txs = ln.split("\t")
            tl = []
            for t in txs.split(";"):
                tn, a, b, c, d, ex = t.split("|")
                tl.append({"name": tn, "tx": [int(a), int(b)], "cds": [int(c), int(d)], "exons": [[int(v) for v in e.split("-")] for e in ex.split(",")]})
            out.append({"name": name, "contig": contig, "start": int(st), "end": int(en), "negative": strand == "-", "transcripts_in_iteration_order": tl})
        return out

    def close(self):
        if getattr(self, "_h", None):
            self._lib.smi_genes_free(self._h)
            self._h = None

    __del__ = close

    def tag(self, ref_id, flags, pos0, cigars):
        """cigars: per record a list of (op letter or code, length).  -> list of (GE, GS, XF); GE / GS None = removed; XF None = untouched"""
        n = len(ref_id)
        rid = np.ascontiguousarray(ref_id, dtype=np.int32)
        fl = np.ascontiguousarray(flags, dtype=np.uint16)
        p0 = np.ascontiguousarray(pos0, dtype=np.int32)
        off = np.zeros(n + 1, dtype=np.uint32)
        flat = []
        for i, cg in enumerate(cigars):
            for op, ln in cg:
                flat.append(int(ln) << 4 | (self.CIGAR_OPS.index(op) if isinstance(op, str) else int(op)))
            off[i + 1] = len(flat)
        cg = np.asarray(flat if flat else [0], dtype=np.uint32)
        return self.tag_arrays(rid, fl, p0, cg, off)

    def tag_arrays(self, rid, fl, p0, cg, off):
        return self._run(self._lib.smi_gene_tag_chunk, (self._h, _ptr(rid), _ptr(fl), _ptr(p0), _ptr(cg), _ptr(off), int(rid.size)), int(rid.size))

    def tag_bam(self, bam, recs):
        """bam: inflated stream (numpy uint8), recs: BAM_RECORD_DTYPE index entries"""
        recs = np.ascontiguousarray(recs)
        return self._run(self._lib.smi_gene_tag_bam, (self._h, _ptr(bam), bam.size, _ptr(recs), int(recs.size)), int(recs.size))

    def tag_bam_raw(self, bam, recs):
        """-> (bytes buffer, uint32 offsets [3 n + 1]): GE, GS, XF of record i at offsets 3 i .. 3 i + 3 (what smi_bam_write_batch takes)"""
        recs = np.ascontiguousarray(recs)
        n = int(recs.size)
        out_off = np.zeros(3 * n + 1, dtype=np.uint32)
        need = ctypes.c_size_t(0)
        args = (self._h, _ptr(bam), bam.size, _ptr(recs), n)
        if self._lib.smi_gene_tag_bam(*args, None, 0, _ptr(out_off), ctypes.byref(need)):
            raise SmiError(self._lib.smi_last_error().decode())
        buf = np.zeros(need.value + 1, dtype=np.uint8)
        if self._lib.smi_gene_tag_bam(*args, _ptr(buf), need.value, _ptr(out_off), ctypes.byref(need)):
            raise SmiError(self._lib.smi_last_error().decode())
        return buf, out_off

    def _run(self, fn, args, n):
        out_off = np.zeros(3 * n + 1, dtype=np.uint32)
        need = ctypes.c_size_t(0)
        if fn(*args, None, 0, _ptr(out_off), ctypes.byref(need)):
            raise SmiError(self._lib.smi_last_error().decode())
        buf = ctypes.create_string_buffer(need.value + 1)
        if fn(*args, buf, need.value, _ptr(out_off), ctypes.byref(need)):
            raise SmiError(self._lib.smi_last_error().decode())
        raw = buf.raw
        res = []
        for i in range(n):
            ge, gs, xf = (raw[out_off[3 * i + k]:out_off[3 * i + k + 1]].decode() for k in range(3))
            res.append((ge or None, gs or None, xf or None))
        return res


class BamWriteConfig(ctypes.Structure):
    _fields_ = [("bc_edit_limit", ctypes.c_int32), ("truncate_read_name", ctypes.c_int32), ("five_prime", ctypes.c_int32), ("n_threads", ctypes.c_int32),
                ("gene_tag", ctypes.c_char * 4)]


def bam_write_bound(recs, batch=None):
    """upper bound of the bytes smi_bam_write_batch writes for these records (every tag this step can add is below 420 bytes per record)"""
    ln = recs["rec_len"] if batch is None else recs["rec_len"][batch]
    return int(ln.sum()) + 420 * int(ln.size) + 64


def bam_write_batch(bam, recs, batch, tags, gene=None, bc_edit_limit=None, truncate_read_name=False, five_prime=False, n_threads=4,
                    gene_counts=None, region=None, nth_record=None, out_bc=None, out_umi=None, gene_tag="GE"):
    """smi_bam_write_batch: the records `batch` (indices into recs) of an inflated BAM -> (bytes of <out>.bam, bytes of <out>_umifound_.bam,
    write order).  tags: UMI_TAG_DTYPE array indexed like recs; gene: (buffer, offsets) of GeneTagger.tag_bam_raw or None; gene_counts: a
    GeneCounts fed in write order (region int64 / nth_record uint8 indexed like recs).  out_bc / out_umi: uint8 arrays to write into (the
    returned arrays are views of them).  gene_tag: -g, the attribute the gene name is written under and counted from (two letters)."""
    lib = load_library()
    recs = np.ascontiguousarray(recs)
    batch = np.ascontiguousarray(batch, dtype=np.int32)
    tags = np.ascontiguousarray(tags, dtype=UMI_TAG_DTYPE)
    cfg = BamWriteConfig()
    lib.smi_bam_write_default_config(ctypes.byref(cfg))
    cfg.bc_edit_limit = -1 if bc_edit_limit is None else int(bc_edit_limit)
    cfg.truncate_read_name, cfg.five_prime, cfg.n_threads = int(bool(truncate_read_name)), int(bool(five_prime)), int(n_threads)
    if len(gene_tag.encode()) > 3:
        raise SmiError("smi_bam_write_batch: the gene name attribute (-g) should have two letters")
    cfg.gene_tag = gene_tag.encode()
    gbuf, goff = (None, None) if gene is None else gene
    if gene_counts is not None:
        region = np.ascontiguousarray(region, dtype=np.int64)
        nth_record = np.ascontiguousarray(nth_record, dtype=np.uint8)
    if out_bc is None:
        cap = bam_write_bound(recs, batch)
        out_bc, out_umi = np.empty(cap, dtype=np.uint8), np.empty(cap, dtype=np.uint8)
    order = np.zeros(max(batch.size, 1), dtype=np.int32)
    nb, nu = ctypes.c_size_t(0), ctypes.c_size_t(0)
    rc = lib.smi_bam_write_batch(_ptr(bam), bam.size, _ptr(recs), _ptr(batch), int(batch.size), _ptr(tags), _ptr(gbuf), _ptr(goff), ctypes.byref(cfg),
                                 _ptr(out_bc), out_bc.size, ctypes.byref(nb), _ptr(out_umi), out_umi.size, ctypes.byref(nu), _ptr(order),
                                 None if gene_counts is None else gene_counts._h, _ptr(region) if gene_counts is not None else None,
                                 _ptr(nth_record) if gene_counts is not None else None)
    if rc:
        raise SmiError(lib.smi_last_error().decode())
    return out_bc[:nb.value], out_umi[:nu.value], order[:batch.size]


def pack_gz_paths(paths, buffer=None):
    """the files of one K-INFLATE round read straight into one host buffer at the 512-byte aligned offsets the kernel wants (1 KiB of zeros
    behind the last) -> dict for Context.gz_inflate_device(packed=...).  buffer: a uint8 array to use (e.g. page-locked, reused from round to
    round) when it is large enough."""
    sizes = [os.path.getsize(p) for p in paths]
    in_off, at = [], 0
    for n in sizes:
        in_off.append(at)
        at = (at + n + 511) & ~511
    total = at + 1024
    host = buffer[:total] if buffer is not None and buffer.size >= total else np.empty(total, dtype=np.uint8)
    out_caps = []
    for p, o, n in zip(paths, in_off, sizes):
        with open(p, "rb") as f:
            got = f.readinto(memoryview(host[o:o + n]))
        if got != n:
            raise SmiError(f"short read of {p}")
        host[o + n:((o + n + 511) & ~511)] = 0
        isize = int.from_bytes(host[o + n - 4:o + n].tobytes(), "little") if n >= 18 else 0
        out_caps.append(isize if isize <= 1032 * n else 0)   # (more than DEFLATE can expand to: not an ISIZE -- capacity 0 sends the file to the host decoder)
    host[at:total] = 0
    return dict(host=host, in_off=in_off, sizes=sizes, out_caps=out_caps)


def bam_name_seen(bam, recs):
    """uint8 per record: a record of the same read name comes earlier (smi_bam_name_seen)"""
    lib = load_library()
    recs = np.ascontiguousarray(recs)
    out = np.zeros(max(int(recs.size), 1), dtype=np.uint8)
    if lib.smi_bam_name_seen(_ptr(bam), bam.size, _ptr(recs), int(recs.size), _ptr(out)):
        raise SmiError(lib.smi_last_error().decode())
    return out


class NameSet:
    """the read names seen so far in a BAM that is read in segments (smi_name_set_*)"""

    def __init__(self):
        self._lib = load_library()
        self._h = ctypes.c_void_p()
        if self._lib.smi_name_set_create(ctypes.byref(self._h)):
            raise SmiError(self._lib.smi_last_error().decode())

    def seen(self, bam, recs, start, nth):
        """nth[start:] of this segment's records filled (1 = the name came earlier), names added"""
        recs = np.ascontiguousarray(recs)
        if self._lib.smi_name_set_seen(self._h, _ptr(bam), bam.size, _ptr(recs), int(start), int(recs.size), _ptr(nth)):
            raise SmiError(self._lib.smi_last_error().decode())

    def close(self):
        if getattr(self, "_h", None):
            self._lib.smi_name_set_free(self._h)
            self._h = None

    __del__ = close


def bam_chunk_inputs(bam, recs, idx):
    """names / CIGARs / flags / positions of records idx in the layout smi_assignumis_chunk takes -> dict of arrays"""
    lib = load_library()
    recs = np.ascontiguousarray(recs)
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    n = int(idx.size)
    a, b = ctypes.c_size_t(0), ctypes.c_size_t(0)
    if lib.smi_bam_chunk_inputs(_ptr(bam), bam.size, _ptr(recs), _ptr(idx), n, None, None, None, None, None, None, ctypes.byref(a), ctypes.byref(b)):
        raise SmiError(lib.smi_last_error().decode())
    names, noff = np.zeros(a.value + 1, dtype=np.uint8), np.zeros(n + 1, dtype=np.uint32)
    cig, coff = np.zeros(b.value + 1, dtype=np.uint32), np.zeros(n + 1, dtype=np.uint32)
    fl, p0 = np.zeros(max(n, 1), dtype=np.uint16), np.zeros(max(n, 1), dtype=np.int32)
    if lib.smi_bam_chunk_inputs(_ptr(bam), bam.size, _ptr(recs), _ptr(idx), n, _ptr(names), _ptr(noff), _ptr(cig), _ptr(coff), _ptr(fl), _ptr(p0),
                                ctypes.byref(a), ctypes.byref(b)):
        raise SmiError(lib.smi_last_error().decode())
    return dict(names=names, name_off=noff, cigars=cig, cigar_off=coff, flags=fl, pos0=p0, n=n)


def two_bit_code(seq):
    """NucleicAcidTwoBitPerBase(String).getSequence() (TB!nuc/encoding/TwoBit/NucleicAcidTwoBitPerBase.java:L183-187, L448-450): two bits per
    base, first base most significant; any other character ORs in the table's fill value -2 like there"""
    v = 0
    for ch in seq:
        c = {"A": 0, "a": 0, "G": 1, "g": 1, "C": 2, "c": 2, "T": 3, "t": 3}.get(ch, -2)
        v = ((v << 2) | (c & 0xFFFFFFFFFFFFFFFF)) & 0xFFFFFFFFFFFFFFFF
    return v


class GeneCounts:
    """The counters behind <out>.genecounts.tsv and <out>.UMIdepths.tsv (smi_gene_counts_*) = GeneCounts
    (FJ!umifinder/scanstats/GeneCounts.java): add() per written batch, the two texts at the end of the run."""

    def __init__(self):
        self._lib = load_library()
        h = ctypes.c_void_p()
        if self._lib.smi_gene_counts_create(ctypes.byref(h)):
            raise SmiError(self._lib.smi_last_error().decode())
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.smi_gene_counts_free(self._h)
            self._h = None

    __del__ = close

    def add(self, gene, region, cell_bc, umi, has_bc_umi, flag, mapq, first_cigar, last_cigar, nth_record, five_prime=False):
        """gene: per record a str or None; region: int (< 0 none); cell_bc / umi: 2-bit codes (two_bit_code); first_cigar / last_cigar:
        `len << 4 | op` (0xFFFFFFFF in first_cigar: no CIGAR)"""
        n = len(region)
        names = [None if g is None else (g.encode() if isinstance(g, str) else g) for g in gene]
        arr = (ctypes.c_char_p * max(n, 1))(*names)
        a = lambda x, t: np.ascontiguousarray(x, dtype=t)  # noqa: E731
        cols = [a(region, np.int64), a(cell_bc, np.uint64), a(umi, np.uint64), a(has_bc_umi, np.uint8), a(flag, np.uint16), a(mapq, np.uint8),
                a(first_cigar, np.uint32), a(last_cigar, np.uint32), a(nth_record, np.uint8)]
        if any(c.size != n for c in cols) or len(names) != n:
            raise SmiError("GeneCounts.add: columns of different lengths")
        if self._lib.smi_gene_counts_add(self._h, n, arr, *[_ptr(c) for c in cols], int(bool(five_prime))):
            raise SmiError(self._lib.smi_last_error().decode())

    def merge(self, other):
        if self._lib.smi_gene_counts_merge(self._h, other._h):
            raise SmiError(self._lib.smi_last_error().decode())

    def dump(self):
        """-> bytes (smi_gene_counts_dump): the tables, to be carried to another process"""
        n = ctypes.c_size_t(0)
        if self._lib.smi_gene_counts_dump(self._h, None, 0, ctypes.byref(n)):
            raise SmiError(self._lib.smi_last_error().decode())
        buf = ctypes.create_string_buffer(max(n.value, 1))
        if self._lib.smi_gene_counts_dump(self._h, buf, n.value, ctypes.byref(n)):
            raise SmiError(self._lib.smi_last_error().decode())
        return buf.raw[:n.value]

    @classmethod
    def load(cls, data):
        self = cls.__new__(cls)
        self._lib = load_library()
        self._h = ctypes.c_void_p()
        b = bytes(data)
        if self._lib.smi_gene_counts_load(b, len(b), ctypes.byref(self._h)):
            raise SmiError(self._lib.smi_last_error().decode())
        return self

    def merge_shard(self, later):
        """fold the tables of a later shard of the same run (assignumis split by chromosome) into this one -> keys whose merge would depend on
        the order of the increments (left alone; 0 for reads whose alignments stay within one shard)"""
        bad = ctypes.c_size_t(0)
        if self._lib.smi_gene_counts_merge_shard(self._h, later._h, ctypes.byref(bad)):
            raise SmiError(self._lib.smi_last_error().decode())
        return int(bad.value)

    def info(self):
        a, b = ctypes.c_int64(0), ctypes.c_int64(0)
        c, d, e = ctypes.c_size_t(0), ctypes.c_size_t(0), ctypes.c_size_t(0)
        self._lib.smi_gene_counts_info(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c), ctypes.byref(d), ctypes.byref(e))
        return dict(records_with_gene=a.value, records_skipped_clipping=b.value, genes=c.value, gene_entries=d.value, region_entries=e.value)

    def _text(self, fn, *args):
        need = ctypes.c_size_t(0)
        if fn(self._h, *args, None, 0, ctypes.byref(need)):
            raise SmiError(self._lib.smi_last_error().decode())
        buf = ctypes.create_string_buffer(need.value + 1)
        if fn(self._h, *args, buf, need.value, ctypes.byref(need)):
            raise SmiError(self._lib.smi_last_error().decode())
        return buf.raw[:need.value].decode()

    def genecounts_tsv(self, bc_length=16):
        return self._text(self._lib.smi_gene_counts_tsv, int(bc_length))

    def umi_depths_tsv(self):
        return self._text(self._lib.smi_umi_depths_tsv)


def format_read_name(read_name, raw_seq, raw_qual, scan, bc=None, rank=0, read_id=0, five_prime=False):
    """smi_format_read_name; scan: SCAN_RESULT_DTYPE record, bc: BC_RESULT_DTYPE record or None -> str"""
    lib = load_library()
    sc = np.zeros(1, dtype=SCAN_RESULT_DTYPE)
    sc[0] = scan
    b = None
    if bc is not None:
        b = np.zeros(1, dtype=BC_RESULT_DTYPE)
        b[0] = bc
    out = ctypes.create_string_buffer(1200)
    n = lib.smi_format_read_name(read_name.encode(), raw_seq.encode(), raw_qual.encode(), len(raw_seq), _ptr(sc),
                                 _ptr(b), int(rank), int(read_id), int(five_prime), out, 1200)
    if n < 0:
        raise SmiError(f"smi_format_read_name error {n}: {lib.smi_last_error().decode()}")
    return out.value.decode()


UMI_ASSIGNMENT_DTYPE = np.dtype([("center", "<i4"), ("offset", "i1"), ("ed", "i1"), ("ed_second", "i1"), ("pos2", "i1")])
UMI_CLUSTER_CONFIG_DTYPE = np.dtype([("complete_link_ed", "<i4"), ("single_link_ed", "<i4"), ("single_link_switch", "<i4"),
                                     ("fold_depth_below_max", "<i4"), ("own_clusterer_above", "<i4")])


def umi_cluster_config(**overrides):
    """smi_umi_cluster_default_config (+ field overrides) -> 1-element structured array"""
    lib = load_library()
    cfg = np.zeros(1, dtype=UMI_CLUSTER_CONFIG_DTYPE)
    if lib.smi_umi_cluster_default_config(_ptr(cfg)) != 0:
        raise SmiError(lib.smi_last_error().decode())
    for k, v in overrides.items():
        cfg[k] = v
    return cfg


def umi_cluster_groups(dist, mat_off, group_off, mean_qv, cfg=None, n_threads=1):
    """smi_umi_cluster_groups on host arrays -> (assignments [n_reads], skipped bool [n_reads])"""
    lib = load_library()
    cfg = umi_cluster_config() if cfg is None else cfg
    dist = np.ascontiguousarray(dist, dtype=np.uint8)
    mat_off = np.ascontiguousarray(mat_off, dtype=np.uint64)
    group_off = np.ascontiguousarray(group_off, dtype=np.uint32)
    qv = np.ascontiguousarray(mean_qv, dtype=np.float32)
    n_reads = int(group_off[-1])
    out = np.zeros(max(n_reads, 1), dtype=UMI_ASSIGNMENT_DTYPE)
    sk = np.zeros(max(n_reads, 1), dtype=np.uint8)
    rc = lib.smi_umi_cluster_groups(_ptr(dist), _ptr(mat_off), _ptr(group_off), group_off.size - 1, _ptr(qv), _ptr(cfg),
                                    _ptr(out), _ptr(sk), int(n_threads))
    if rc != 0:
        raise SmiError(f"smi_umi_cluster_groups error {rc}: {lib.smi_last_error().decode()}")
    return out[:n_reads], sk[:n_reads].astype(bool)


def region_group(pos, reverse, max_dist=500, keep_data_end=False):
    """smi_region_group; pos: sequence with None for reads without a clustering position -> (region list, n_done)"""
    lib = load_library()
    n = len(pos)
    p = np.array([0 if v is None else v for v in pos], dtype=np.int32)
    h = np.array([v is not None for v in pos], dtype=np.uint8)
    r = np.ascontiguousarray(np.asarray(reverse, dtype=bool), dtype=np.uint8)
    out = np.zeros(max(n, 1), dtype=np.int32)
    nd = ctypes.c_int32(0)
    rc = lib.smi_region_group(_ptr(p), _ptr(h), _ptr(r), n, int(max_dist), int(keep_data_end), _ptr(out), ctypes.byref(nd))
    if rc != 0:
        raise SmiError(f"smi_region_group error {rc}: {lib.smi_last_error().decode()}")
    return out[:n].tolist(), nd.value


def ref_position_at_read_position(cigar, alignment_start, position):
    """cigar: [(op char, length)] -> reference position or None"""
    lib = load_library()
    c = np.array([(ln << 4) | "MIDNSHP=X".index(op) for op, ln in cigar], dtype=np.uint32)
    out = ctypes.c_int32(0)
    rc = lib.smi_ref_position_at_read_position(_ptr(c), c.size, int(alignment_start), int(position), ctypes.byref(out))
    if rc < 0:
        raise SmiError(f"smi_ref_position_at_read_position error {rc}: {lib.smi_last_error().decode()}")
    return out.value if rc == 1 else None


def ref_position_at_read_position_raw(cigar_u32, alignment_start, position):
    """same on a BAM-encoded CIGAR (numpy uint32: len << 4 | op)"""
    lib = load_library()
    c = np.ascontiguousarray(cigar_u32, dtype=np.uint32)
    out = ctypes.c_int32(0)
    rc = lib.smi_ref_position_at_read_position(_ptr(c), c.size, int(alignment_start), int(position), ctypes.byref(out))
    if rc < 0:
        raise SmiError(f"smi_ref_position_at_read_position error {rc}: {lib.smi_last_error().decode()}")
    return out.value if rc == 1 else None


UMI_TAG_DTYPE = np.dtype([("region", "<i4"), ("center", "<i4"), ("u1", "i1"), ("u2", "i1"), ("flags", "u1"), ("reserved", "u1"),
                          ("u8", "S12"), ("u7", "S12")])
UMI_HAS_BC, UMI_HAS_U7, UMI_CLUSTERED, UMI_SKIPPED = 1, 2, 4, 8


class AssignUmisConfig(ctypes.Structure):
    """smi_assignumis_config"""
    _fields_ = [("max_dist", ctypes.c_int32), ("grouping_distance", ctypes.c_int32), ("bc_edit_limit", ctypes.c_int32),
                ("keep_data_end", ctypes.c_int32), ("n_threads", ctypes.c_int32), ("five_prime", ctypes.c_int32),
                ("umi_length", ctypes.c_int32), ("cluster", ctypes.c_void_p), ("random_umi_seed", ctypes.c_uint64)]


class PinnedBuffer:
    """page-locked host bytes (smi_host_alloc) as a numpy uint8 array: .array; free with .close()"""

    def __init__(self, n_bytes):
        lib = load_library()
        self._p = ctypes.c_void_p(0)
        if lib.smi_host_alloc(int(n_bytes), ctypes.byref(self._p)):
            raise SmiError(lib.smi_last_error().decode())
        self.array = np.ctypeslib.as_array(ctypes.cast(self._p, ctypes.POINTER(ctypes.c_uint8)), shape=(int(n_bytes),))

    def close(self):
        if self._p:
            load_library().smi_host_free(self._p)
            self._p, self.array = ctypes.c_void_p(0), None


FASTQ_RECORD_DTYPE = np.dtype([("name_start", "<u8"), ("seq_start", "<u8"), ("plus_start", "<u8"), ("qual_start", "<u8"),
                               ("name_len", "<u4"), ("seq_len", "<u4"), ("plus_len", "<u4"), ("reserved", "<u4")])
assert FASTQ_RECORD_DTYPE.itemsize == 48


N_READ_FLAGS = 37
N_SCAN_STATS = N_READ_FLAGS + 3   # smi_scan_stats as a uint64 vector: the 37 counters, sum_len_passed, sum_len_failed, n_reads_split
READ_FLAG_NAMES = ["ALL_READS", "CHIMERIC_READS_SPLIT", "MULTI_CHIMERIC_READS_DISCARDED", "ALL_READS_AFTER_SPLIT", "READS_AFTER_SPLIT", "PASSED_TOTAL", "FAILED",
                   "MEAN_LENGTH_PASSED", "MEAN_LENGTH_FAILED", "PASSED_FWD", "PASSED_REV", "PASSED_TOT_TSO", "POLY_T_5P", "POLY_A_3P", "POLY_A_NOT_FOUND",
                   "POLY_T_5P_POLY_A_3P", "ADAPTER_5P", "ADAPTER_3P", "TSO_5P", "TSO_3P", "ADAPTER_SELECTED_DESP_ADAPTER_BOTH_SIDES", "READ_TOO_SHORT",
                   "ADAPTER_5P_AND_3P", "TSO_5P_AND_3P", "TSO_5P_AND_3P_FAILED", "BC_FOUND", "BC_FOUND_NO_SECONDARY_MATCH", "BC_FOUND_ED0", "BC_FOUND_ED1",
                   "BC_FOUND_ED2", "BC_FOUND_ED3", "BC_ED_DIFF_ABOVE2", "BC_ED_DIFF1", "BC_ED_DIFF2", "BC_OFFSET0", "BC_OFFSET1", "BC_OFFSET2"]


def record_flags(scan, bc=None, from_split=False, multi_chimeric=False):
    """smi_record_flags: the reference's 64-bit flag word of one record (SCAN_RESULT_DTYPE / BC_RESULT_DTYPE records)"""
    lib = load_library()
    sc = np.zeros(1, dtype=SCAN_RESULT_DTYPE)
    sc[0] = scan
    b = None
    if bc is not None:
        b = np.zeros(1, dtype=BC_RESULT_DTYPE)
        b[0] = bc
    return int(lib.smi_record_flags(_ptr(sc), _ptr(b), int(from_split), int(multi_chimeric)))


def scan_stats_tsv(stats):
    """smi_scan_stats_tsv: the text of ReadFlags.print from a smi_scan_stats vector (uint64 [40]; vectors of several chunks / runs add up)"""
    lib = load_library()
    st = np.ascontiguousarray(stats, dtype=np.uint64)
    assert st.size == N_SCAN_STATS
    n = ctypes.c_size_t(0)
    if lib.smi_scan_stats_tsv(st.ctypes.data, None, 0, ctypes.byref(n)):
        raise SmiError(lib.smi_last_error().decode())
    out = ctypes.create_string_buffer(n.value + 1)
    if lib.smi_scan_stats_tsv(st.ctypes.data, out, n.value, ctypes.byref(n)):
        raise SmiError(lib.smi_last_error().decode())
    return out.raw[:n.value].decode()


class Pass2Decisions(ctypes.Structure):
    """smi_pass2_decisions"""
    _fields_ = [("n_records_in", ctypes.c_size_t), ("n_records_out", ctypes.c_size_t), ("chim", ctypes.c_void_p),
                ("frag_offsets", ctypes.c_void_p), ("frag_src", ctypes.c_void_p), ("scan", ctypes.c_void_p), ("bc", ctypes.c_void_p),
                ("rank", ctypes.c_void_p)]


class PackedReads(ctypes.Structure):
    """smi_packed_reads"""
    _fields_ = [("planes", ctypes.c_void_p), ("stride", ctypes.c_size_t), ("pstart", ctypes.c_void_p), ("n_seg", ctypes.c_int32),
                ("reserved", ctypes.c_int32), ("total_words", ctypes.c_size_t), ("seg_host_word", ctypes.c_uint64 * 256),
                ("seg_dev_word", ctypes.c_uint64 * 256), ("seg_words", ctypes.c_uint64 * 256)]


def fastq_index_pack_host(text, n_threads=1, cap_records=None):
    """smi_fastq_index_pack_host -> (recs, offsets, error bits, pstart or None, planes uint32 (host layout), PackedReads)"""
    lib = load_library()
    buf = _as_u8(text)
    cap = int(cap_records) if cap_records is not None else buf.size // 6 + 2
    recs = np.zeros(cap + 1, dtype=FASTQ_RECORD_DTYPE)
    offs = np.zeros(cap + 2, dtype=np.uint64)
    pstart = np.zeros(cap + 1, dtype=np.uint32)
    words = max(int(lib.smi_packed_planes_words(buf.size, int(n_threads))), read_planes_words(buf.size // 2, cap))
    planes = np.full(words, 0xDEADBEEF, dtype=np.uint32)
    pk = PackedReads()
    n, err = ctypes.c_size_t(0), ctypes.c_uint32(0)
    rc = lib.smi_fastq_index_pack_host(buf.ctypes.data if buf.size else None, buf.size, recs.ctypes.data, offs.ctypes.data, pstart.ctypes.data, cap,
                                       planes.ctypes.data, planes.size, ctypes.byref(pk), ctypes.byref(n), ctypes.byref(err), int(n_threads))
    if rc != 0:
        raise SmiError(f"smi_fastq_index_pack_host error {rc}: {lib.smi_last_error().decode()}")
    return recs[:n.value], offs[:n.value + 1], err.value, (pstart[:n.value] if pk.pstart else None), planes, pk


def _as_u8(text):
    return text if isinstance(text, np.ndarray) else np.frombuffer(text, dtype=np.uint8)


def fastq_index_host(text, n_threads=1, cap_records=None):
    """smi_fastq_index_host -> (FASTQ_RECORD_DTYPE [n], offsets uint64 [n + 1], error bits)"""
    lib = load_library()
    buf = _as_u8(text)
    cap = int(cap_records) if cap_records is not None else buf.size // 6 + 2
    recs = np.zeros(cap + 1, dtype=FASTQ_RECORD_DTYPE)
    offs = np.zeros(cap + 2, dtype=np.uint64)
    n, err = ctypes.c_size_t(0), ctypes.c_uint32(0)
    rc = lib.smi_fastq_index_host(buf.ctypes.data if buf.size else None, buf.size, recs.ctypes.data, offs.ctypes.data, cap, ctypes.byref(n),
                                  ctypes.byref(err), int(n_threads))
    if rc != 0:
        raise SmiError(f"smi_fastq_index_host error {rc}: {lib.smi_last_error().decode()}")
    return recs[:n.value], offs[:n.value + 1], err.value


def read_planes_words(total_bases, n):
    return int(load_library().smi_read_planes_words(int(total_bases), int(n)))


def pack_reads_host(text, recs, offsets, n_threads=1):
    """smi_pack_reads_host -> planes, uint32 [smi_read_planes_words(offsets[-1], n)]"""
    lib = load_library()
    buf = _as_u8(text)
    n = int(recs.size)
    recs = np.ascontiguousarray(recs, dtype=FASTQ_RECORD_DTYPE)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    planes = np.full(read_planes_words(int(offsets[n]) if n else 0, n), 0xDEADBEEF, dtype=np.uint32)  # every word must be overwritten
    if lib.smi_pack_reads_host(buf.ctypes.data, recs.ctypes.data, offsets.ctypes.data, n, planes.ctypes.data, int(n_threads)):
        raise SmiError(lib.smi_last_error().decode())
    return planes


def pack_quals_host(text, recs, five_prime=False, n_threads=1):
    """smi_pack_quals_host -> (qtail uint8 [n, 224], qsum uint32 [n])"""
    lib = load_library()
    buf = _as_u8(text)
    n = int(recs.size)
    recs = np.ascontiguousarray(recs, dtype=FASTQ_RECORD_DTYPE)
    qtail = np.zeros((max(n, 1), END_BASES), dtype=np.uint8)
    qsum = np.zeros(max(n, 1), dtype=np.uint32)
    if lib.smi_pack_quals_host(buf.ctypes.data, recs.ctypes.data, n, int(bool(five_prime)), qtail.ctypes.data, qsum.ctypes.data, int(n_threads)):
        raise SmiError(lib.smi_last_error().decode())
    return qtail[:n], qsum[:n]


def fastq_write_host(text, recs, offsets, scan, bc, frag_offsets=None, frag_src=None, chim=None, rank=None, first_read_id=1,
                     five_prime=False, trim_fastq=False, n_threads=1, cap=None):
    """smi_fastq_write_host on numpy decisions (SCAN_RESULT_DTYPE / BC_RESULT_DTYPE per output record; frag_offsets / frag_src / chim as
    smi_split_offsets_device gives them, or all None when the splitter did not run) -> (passed bytes, failed bytes, records passed)"""
    lib = load_library()
    buf = _as_u8(text)
    recs = np.ascontiguousarray(recs, dtype=FASTQ_RECORD_DTYPE)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    scan = np.ascontiguousarray(scan, dtype=SCAN_RESULT_DTYPE)
    bc = np.ascontiguousarray(bc, dtype=BC_RESULT_DTYPE)
    m = int(scan.size)
    keep = [buf, recs, offsets, scan, bc]
    d = Pass2Decisions()
    d.n_records_in, d.n_records_out = int(recs.size), m
    d.scan, d.bc = scan.ctypes.data, bc.ctypes.data
    if frag_src is not None:
        fo = np.ascontiguousarray(frag_offsets, dtype=np.uint64)
        fs = np.ascontiguousarray(frag_src, dtype=np.uint32)
        ch = np.ascontiguousarray(chim, dtype=CHIMERA_RESULT_DTYPE)
        keep += [fo, fs, ch]
        d.frag_offsets, d.frag_src, d.chim = fo.ctypes.data, fs.ctypes.data, ch.ctypes.data
    else:
        d.frag_offsets = offsets.ctypes.data
    if rank is not None:
        rk = np.ascontiguousarray(rank, dtype=np.int32)
        keep.append(rk)
        d.rank = rk.ctypes.data
    cap = int(cap) if cap is not None else 3 * buf.size + 400 * m + 64
    out_p, out_f = np.empty(cap, dtype=np.uint8), np.empty(cap, dtype=np.uint8)
    cfg = (ctypes.c_int32 * 2)(int(bool(five_prime)), int(bool(trim_fastq)))
    totals = (ctypes.c_uint64 * 3)()
    err = ctypes.c_uint32(0)
    rc = lib.smi_fastq_write_host(buf.ctypes.data, recs.ctypes.data, offsets.ctypes.data, ctypes.byref(d), int(first_read_id), ctypes.byref(cfg),
                                  out_p.ctypes.data, cap, out_f.ctypes.data, cap, totals, ctypes.byref(err), int(n_threads))
    if rc != 0:
        raise SmiError(f"smi_fastq_write_host: {lib.smi_last_error().decode()} (error bits {err.value})")
    return out_p[:totals[0]].tobytes(), out_f[:totals[1]].tobytes(), int(totals[2])


class DeviceSpan:
    """bytes in a context's device memory that the library still owns (smi_pass2_config.device_output): address and size; tensor() wraps them as a
    uint8 torch view without copying (building the view costs more than the pointer, so it is made on request)"""
    __slots__ = ("ptr", "nbytes", "device")

    def __init__(self, ptr, nbytes, device):
        self.ptr, self.nbytes, self.device = ptr, nbytes, device

    def __len__(self):
        return self.nbytes

    @property
    def __cuda_array_interface__(self):
        return dict(shape=(self.nbytes,), typestr="|u1", data=(self.ptr, False), version=2)

    def tensor(self):
        import torch

        dev = torch.device("cuda", self.device)
        return torch.as_tensor(self, device=dev) if self.nbytes else torch.empty(0, dtype=torch.uint8, device=dev)

    def tobytes(self):
        return bytes(self.tensor().cpu().numpy())


class Pass2Config(ctypes.Structure):
    """smi_pass2_config"""
    _fields_ = [("max_ed", ctypes.c_int32), ("five_prime", ctypes.c_int32), ("dont_search_polya", ctypes.c_int32),
                ("split_chimeras", ctypes.c_int32), ("trim_fastq", ctypes.c_int32), ("want_results", ctypes.c_int32),
                ("first_read_id", ctypes.c_uint32), ("compress", ctypes.c_uint32), ("rank_keys", ctypes.c_void_p),
                ("rank_values", ctypes.c_void_p), ("n_ranks", ctypes.c_size_t), ("device_output", ctypes.c_uint32), ("reserved", ctypes.c_uint32)]


class Pass2Output(ctypes.Structure):
    """smi_pass2_output"""
    _fields_ = [("passed", ctypes.c_void_p), ("failed", ctypes.c_void_p), ("passed_bytes", ctypes.c_size_t),
                ("failed_bytes", ctypes.c_size_t), ("n_records_in", ctypes.c_size_t), ("n_records_out", ctypes.c_size_t),
                ("n_passed", ctypes.c_size_t), ("scan", ctypes.c_void_p), ("bc", ctypes.c_void_p), ("fastq_errors", ctypes.c_uint32),
                ("reserved", ctypes.c_uint32), ("stats", ctypes.c_void_p), ("passed_text_bytes", ctypes.c_size_t),
                ("failed_text_bytes", ctypes.c_size_t)]


class ChimeraConfig(ctypes.Structure):
    """smi_chimera_config"""
    _fields_ = [("tso_complete", ctypes.c_char_p), ("adapter_complete", ctypes.c_char_p), ("tso_max_errors", ctypes.c_int32),
                ("adapter_max_errors", ctypes.c_int32), ("internal_pat_len", ctypes.c_int32),
                ("internal_pat_frac", ctypes.c_float), ("window_polya", ctypes.c_int32), ("bc_umi_len", ctypes.c_int32)]


def chimera_fragment_name(read_name, result, fragment):
    """smi_chimera_fragment_name; result: CHIMERA_RESULT_DTYPE record -> str"""
    lib = load_library()
    r = np.zeros(1, dtype=CHIMERA_RESULT_DTYPE)
    r[0] = result
    out = ctypes.create_string_buffer(len(read_name) + 64)
    n = lib.smi_chimera_fragment_name(read_name.encode(), _ptr(r), int(fragment), out, len(read_name) + 64)
    if n < 0:
        raise SmiError(f"smi_chimera_fragment_name error {n}: {lib.smi_last_error().decode()}")
    return out.value.decode()


BAM_RECORD_DTYPE = np.dtype([("rec_off", "<u8"), ("name_off", "<u8"), ("cigar_off", "<u8"), ("seq_off", "<u8"), ("qual_off", "<u8"),
                             ("aux_off", "<u8"), ("rec_len", "<u4"), ("aux_len", "<u4"), ("ref_id", "<i4"), ("pos", "<i4"),
                             ("l_seq", "<i4"), ("next_ref_id", "<i4"), ("next_pos", "<i4"), ("tlen", "<i4"), ("flag", "<u2"),
                             ("n_cigar", "<u2"), ("mapq", "u1"), ("l_read_name", "u1"), ("reserved", "u1", (2,))])


def bgzf_inflate(data, n_threads=4, front=None, room=0):
    """inflated bytes of the complete BGZF blocks of `data` (numpy uint8) -> (numpy uint8, bytes consumed).  front: bytes to put in front of
    the inflated data in the SAME buffer (a stream's pending tail) -- the blocks are inflated straight behind them.  room > 0: that many
    spare bytes are left in front instead and (whole buffer, offset of the inflated data, bytes consumed) is returned, for a reader thread
    that inflates ahead of the consumer who knows the pending bytes."""
    lib = load_library()
    data = np.ascontiguousarray(data, dtype=np.uint8)
    n_out, n_blk, used = ctypes.c_size_t(0), ctypes.c_size_t(0), ctypes.c_size_t(0)
    if lib.smi_bgzf_uncompressed_size(data.ctypes.data, data.size, ctypes.byref(n_out), ctypes.byref(n_blk), ctypes.byref(used)):
        raise SmiError(lib.smi_last_error().decode())
    n_front = int(room) if room else (0 if front is None else int(front.size))
    out = np.empty(max(n_front + n_out.value, 1), dtype=np.uint8)
    if n_front and not room:
        out[:n_front] = front
    got = ctypes.c_size_t(0)
    if lib.smi_bgzf_inflate(data.ctypes.data, data.size, out[n_front:].ctypes.data, out.size - n_front, ctypes.byref(got), ctypes.byref(used),
                            int(n_threads)):
        raise SmiError(lib.smi_last_error().decode())
    if room:
        return out[:n_front + got.value], n_front, used.value
    return out[:n_front + got.value], used.value


def bgzf_deflate(data, level=5, block_bytes=0xFF00, n_threads=4):
    """BGZF-compressed bytes (numpy uint8) of `data`, EOF block included"""
    lib = load_library()
    data = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8) if isinstance(data, (bytes, bytearray)) else data, dtype=np.uint8)
    n = ctypes.c_size_t(0)
    if lib.smi_bgzf_deflate(data.ctypes.data, data.size, None, 0, ctypes.byref(n), int(level), int(block_bytes), int(n_threads)):
        raise SmiError(lib.smi_last_error().decode())
    out = np.empty(n.value, dtype=np.uint8)
    if lib.smi_bgzf_deflate(data.ctypes.data, data.size, out.ctypes.data, out.size, ctypes.byref(n), int(level), int(block_bytes),
                            int(n_threads)):
        raise SmiError(lib.smi_last_error().decode())
    return out[:n.value]


def assigned_tsv(keys, counts, max_ed=1):
    """BarcodesAssigned.tsv text from the per-(barcode, ed) counters of smi_bc_counts_device (counts: [n_keys, 3] uint32)"""
    lib = load_library()
    k = np.ascontiguousarray(keys, dtype=np.uint64)
    c = np.ascontiguousarray(counts, dtype=np.uint32).reshape(-1)
    n = ctypes.c_size_t(0)
    if lib.smi_assigned_tsv(k.ctypes.data, c.ctypes.data, k.size, int(max_ed), None, 0, ctypes.byref(n)):
        raise SmiError(lib.smi_last_error().decode())
    out = ctypes.create_string_buffer(n.value + 1)
    if lib.smi_assigned_tsv(k.ctypes.data, c.ctypes.data, k.size, int(max_ed), out, n.value, ctypes.byref(n)):
        raise SmiError(lib.smi_last_error().decode())
    return out.raw[:n.value].decode()


def gz_inflate(data, alloc=None):
    """inflated bytes of a (multi-member) gzip stream, numpy uint8 in and out (smi_gz_inflate_into: the library's own decoder on the calling
    thread; the GIL is released meanwhile).  alloc(n) -> (uint8 array of n bytes, owner): where the text should land (e.g. page-locked
    memory); returns (text, owner) then."""
    lib = load_library()
    data = np.ascontiguousarray(data, dtype=np.uint8)
    make = alloc or (lambda n: (np.empty(n, dtype=np.uint8), None))
    # a single-member file says its size (mod 2^32) in its last four bytes; a file of several members gets room for the usual ratio first
    hint = int.from_bytes(data[-4:].tobytes(), "little") if data.size >= 18 else 0
    if hint > 1032 * data.size:          # more than DEFLATE can expand to: the last bytes are not a member's ISIZE (padding behind the last member)
        hint = 0
    cap = max(hint if hint >= data.size // 2 else 4 * data.size, 64)
    out, owner = make(cap)
    ip, op = ctypes.c_size_t(0), ctypes.c_size_t(0)
    while True:
        rc = lib.smi_gz_inflate_into(data.ctypes.data, data.size, ctypes.byref(ip), out.ctypes.data, out.size, ctypes.byref(op))
        if rc == 0:
            break
        if rc != 1:
            raise SmiError(lib.smi_last_error().decode())
        bigger, owner2 = make(2 * out.size + (64 << 10))
        bigger[:op.value] = out[:op.value]
        if owner is not None:
            owner.close()
        out, owner = bigger, owner2
    return out[:op.value] if alloc is None else (out[:op.value], owner)


def bam_header(bam):
    """-> (header text, [(reference name, length)], offset of the first record)"""
    lib = load_library()
    t_off, t_len, n_ref, rec = ctypes.c_uint64(0), ctypes.c_uint32(0), ctypes.c_int32(0), ctypes.c_uint64(0)
    if lib.smi_bam_header(bam.ctypes.data, bam.size, ctypes.byref(t_off), ctypes.byref(t_len), ctypes.byref(n_ref), None, None,
                          None, 0, ctypes.byref(rec)):
        raise SmiError(lib.smi_last_error().decode())
    k = max(n_ref.value, 1)
    no, nl, rl = np.zeros(k, dtype=np.uint64), np.zeros(k, dtype=np.uint32), np.zeros(k, dtype=np.int32)
    if lib.smi_bam_header(bam.ctypes.data, bam.size, ctypes.byref(t_off), ctypes.byref(t_len), ctypes.byref(n_ref), no.ctypes.data,
                          nl.ctypes.data, rl.ctypes.data, k, ctypes.byref(rec)):
        raise SmiError(lib.smi_last_error().decode())
    text = bam[t_off.value:t_off.value + t_len.value].tobytes().decode(errors="replace")
    # (names as htsjdk reads them: one char per byte -- a damaged name is still a name, not a decoding error)
    refs = [(bam[int(no[i]):int(no[i]) + int(nl[i])].tobytes().decode("latin-1"), int(rl[i])) for i in range(n_ref.value)]
    return text, refs, rec.value


def bam_index_records(bam, start, cap):
    """-> (BAM_RECORD_DTYPE array, offset behind the last complete record)"""
    lib = load_library()
    recs = np.zeros(max(cap, 1), dtype=BAM_RECORD_DTYPE)
    n, end = ctypes.c_size_t(0), ctypes.c_uint64(0)
    if lib.smi_bam_index_records(bam.ctypes.data, bam.size, int(start), recs.ctypes.data, int(cap), ctypes.byref(n), ctypes.byref(end)):
        raise SmiError(lib.smi_last_error().decode())
    return recs[:n.value], end.value


class RunKnobs(ctypes.Structure):
    """smi_run_knobs: the knobs of Jar/config.xml the library takes at run time (include/sicelore_mi.h)"""
    _fields_ = [("min_read_length", ctypes.c_int32), ("min_mean_bc_qv", ctypes.c_int32), ("min_mean_read_qv", ctypes.c_int32),
                ("min_adapter_3p_matches", ctypes.c_int32), ("polya_len", ctypes.c_int32), ("polya_frac", ctypes.c_float),
                ("window_polya", ctypes.c_int32), ("internal_pat_len", ctypes.c_int32), ("internal_pat_frac", ctypes.c_float),
                ("adapter3p", ctypes.c_char * 32), ("adapter3p_complete", ctypes.c_char * 32), ("adapter3p_max_mm", ctypes.c_int32),
                ("adapter3p_complete_max_mm", ctypes.c_int32), ("adapter5p", ctypes.c_char * 32), ("adapter5p_complete", ctypes.c_char * 32),
                ("adapter5p_max_mm", ctypes.c_int32), ("adapter5p_complete_max_mm", ctypes.c_int32), ("adapter5p_window", ctypes.c_int32),
                ("adapter3p5_complete", ctypes.c_char * 32), ("adapter3p5_complete_max_mm", ctypes.c_int32), ("tso_complete", ctypes.c_char * 32),
                ("tso_complete_max_mm", ctypes.c_int32), ("umi_length", ctypes.c_int32), ("tso_scan", ctypes.c_char * 20),
                ("tso_scan_max_mm", ctypes.c_int32), ("tso_scan_min_consec", ctypes.c_int32), ("tso_scan_min_two_best", ctypes.c_int32),
                ("tso_scan_window", ctypes.c_int32), ("reserved", ctypes.c_int32 * 6)]

    def as_dict(self):
        out = {}
        for name, _ in self._fields_[:-1]:
            v = getattr(self, name)
            out[name] = v.decode() if isinstance(v, bytes) else v
        return out


# config.xml element ("section/knob" under <Parameters>) -> field of smi_run_knobs
KNOB_FIELDS = {
    "readscanner/minReadLength": "min_read_length", "readscanner/minMeanBCqv": "min_mean_bc_qv", "readscanner/minMeanReadqv": "min_mean_read_qv",
    "readscanner/minAdapter3pMatches": "min_adapter_3p_matches",
    "polyAT/polyATlength": "polya_len", "polyAT/fractionATInPolyAT": "polya_frac", "polyAT/windowSearchForPolyA": "window_polya",
    "polyAT/internalpATlength": "internal_pat_len", "polyAT/internalFractionATInPolyAT": "internal_pat_frac",
    "adapter_for3pBarcoding/sequence": "adapter3p", "adapter_for3pBarcoding/sequence_complete": "adapter3p_complete",
    "adapter_for3pBarcoding/maxNeedlemanMismatches": "adapter3p_max_mm", "adapter_for3pBarcoding/maxCompleteSeqNeedlemanMismatches": "adapter3p_complete_max_mm",
    "fiveprimeadapter_for5pBarcoding/sequence": "adapter5p", "fiveprimeadapter_for5pBarcoding/sequence_complete": "adapter5p_complete",
    "fiveprimeadapter_for5pBarcoding/maxNeedlemanMismatches": "adapter5p_max_mm",
    "fiveprimeadapter_for5pBarcoding/maxCompleteSeqNeedlemanMismatches": "adapter5p_complete_max_mm",
    "fiveprimeadapter_for5pBarcoding/AdapterSearchWindow": "adapter5p_window",
    "threeprimeadapter_for5pBarcoding/sequence_complete": "adapter3p5_complete",
    "threeprimeadapter_for5pBarcoding/maxCompleteSeqNeedlemanMismatches": "adapter3p5_complete_max_mm",
    "tso_for3pBarcoding/sequence_complete": "tso_complete", "tso_for3pBarcoding/maxCompleteSeqNeedlemanMismatches": "tso_complete_max_mm",
    "umis/umi_length": "umi_length",
    "tso_for3pBarcoding/sequence": "tso_scan", "tso_for3pBarcoding/maxNeedlemanMismatches": "tso_scan_max_mm",
    "tso_for3pBarcoding/minTSO_NeedlemanConsecutiveMatches": "tso_scan_min_consec",
    "tso_for3pBarcoding/minTSO_TwoBestConsecutiveMatches": "tso_scan_min_two_best", "tso_for3pBarcoding/windowForTSOsearch": "tso_scan_window",
}


def run_knobs(**overrides):
    """smi_run_knobs with the shipped config.xml values, fields replaced by name (RunKnobs) or by config.xml element (KNOB_FIELDS)"""
    k = RunKnobs()
    lib = load_library()
    if lib.smi_run_knobs_default(ctypes.byref(k)):
        raise SmiError(lib.smi_last_error().decode())
    kinds = dict(RunKnobs._fields_)
    for name, v in overrides.items():
        f = KNOB_FIELDS.get(name, name)
        if f not in kinds or f == "reserved":
            raise SmiError(f"no run-time knob {name!r}")
        t = kinds[f]
        if t is ctypes.c_float or t is ctypes.c_int32:
            try:
                setattr(k, f, float(v) if t is ctypes.c_float else int(v))
            except ValueError:
                raise SmiError(f"{name} = {v!r}: not a number")
        else:
            b = v.encode() if isinstance(v, str) else bytes(v)
            if len(b) > ctypes.sizeof(t) - 1:
                raise SmiError(f"{name}: sequence of {len(b)} bases (this build: up to {27 if ctypes.sizeof(t) == 32 else 16} here)")
            setattr(k, f, b)
    return k


def check_run_knobs(knobs):
    """the limits of this build, without a device (what smi_ctx_set_knobs checks): SmiError names the knob"""
    lib = load_library()
    cfg = np.zeros(1, dtype=SCAN_CONFIG_DTYPE)
    if lib.smi_scan_config_from_knobs(ctypes.byref(knobs), 2, 0, 0, _ptr(cfg)):
        raise SmiError(lib.smi_last_error().decode())


class Context:
    """One per GPU.  Stands where the reference keeps ``hashMapForBCfinding`` + a ``Parser`` worker
    (FJ!nanoporereadscanner/analyzers/Parser.java:L70-78)."""

    def __init__(self, device=0, _lane_of=None):
        self._lib = load_library()
        h = ctypes.c_void_p()
        if _lane_of is None:
            self._check(self._lib.smi_ctx_create(int(device), ctypes.byref(h)))
        else:
            self._check(self._lib.smi_ctx_create_lane(_lane_of._h, ctypes.byref(h)))
        self._h = h
        self.device = int(device)
        self.n_keys = 0 if _lane_of is None else _lane_of.n_keys
        self._owner = _lane_of

    def lane(self):
        """a worker lane of this context (smi_ctx_create_lane): own stream / arena / pinned buffers, this context's barcode set"""
        return Context(self.device, _lane_of=self)

    def set_polya(self, polya_len=0, polya_frac=0.0, window_polya=0):
        """-p / -f / -w of scanfastq for this context's chunk workers (smi_ctx_set_polya; 0 keeps the shipped value); lanes created or
        refreshed afterwards take them over"""
        self._check(self._lib.smi_ctx_set_polya(self._h, int(polya_len), float(polya_frac), int(window_polya)))

    def set_random_barcodes(self, seed=0):
        """scanfastq -e / --randomBarcode for this context's pass-2 chunk workers (smi_ctx_set_random_barcodes; 0 = off)"""
        self._check(self._lib.smi_ctx_set_random_barcodes(self._h, int(seed)))

    def set_knobs(self, knobs=None):
        """config.xml's knobs for this context's chunk workers (smi_ctx_set_knobs; None: the shipped file); lanes created or refreshed
        afterwards take them over.  SmiError names the knob this build has no kernel for."""
        self._check(self._lib.smi_ctx_set_knobs(self._h, ctypes.byref(knobs) if knobs is not None else None))

    def get_knobs(self):
        k = RunKnobs()
        self._check(self._lib.smi_ctx_get_knobs(self._h, ctypes.byref(k)))
        return k

    def refresh(self):
        """lane: pick up the barcode set its owner has loaded since (smi_ctx_lane_refresh)"""
        self._check(self._lib.smi_ctx_lane_refresh(self._h))
        self.n_keys = self._owner.n_keys

    def _check(self, rc):
        if rc != 0:
            raise SmiError(f"libsicelore_mi error {rc}: {self._lib.smi_last_error().decode()}")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.smi_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- barcode set -------------------------------------------------------------------------------------
    def set_barcode_set(self, keys, mode=SET_USED_LIST):
        """keys: iterable of 2-bit packed 16-mers (host).  Builds the device membership pyramid."""
        k = np.ascontiguousarray(np.asarray(keys, dtype=np.uint64))
        self._check(self._lib.smi_set_barcode_set(self._h, _ptr(k), k.size, int(mode)))
        self.n_keys = int(np.unique(k).size)

    def set_stats(self, digests=False):
        """smi_set_stats: what the last set_barcode_set built -> dict(keys, hbm_bytes, build_ms[, nb_bits, nb5_bits, nb5_digest, nt_slots, nt_entries])"""
        out = np.zeros(8, dtype=np.uint64)
        self._check(self._lib.smi_set_stats(self._h, out.ctypes.data, int(bool(digests))))
        d = dict(keys=int(out[0]), hbm_bytes=int(out[1]), build_ms=float(out[2]) / 1e3)
        if digests:
            d.update(nb_bits=int(out[3]), nb5_bits=int(out[4]), nb5_digest=int(out[5]), nt_slots=int(out[6]), nt_entries=int(out[7]))
        return d

    def set_barcode_set_device(self, d_keys_u32, mode=SET_USED_LIST, stream=None):
        """d_keys_u32: device tensor of int32/uint32 keys."""
        self._check(self._lib.smi_set_barcode_set_device(self._h, _ptr(d_keys_u32), d_keys_u32.numel(), int(mode),
                                                         _stream_ptr(stream)))

    # ---- matcher -----------------------------------------------------------------------------------------
    def bc_match(self, windows, max_ed=1, five_prime=False):
        """Host buffers in, host buffers out (numpy structured arrays)."""
        w = np.ascontiguousarray(windows, dtype=BC_WINDOW_DTYPE)
        out = np.zeros(w.size, dtype=BC_RESULT_DTYPE)
        self._check(self._lib.smi_bc_match_batch(self._h, _ptr(w), w.size, int(max_ed), int(bool(five_prime)),
                                                 _ptr(out)))
        return out

    def bc_match_device(self, d_windows, d_out, n, max_ed=1, five_prime=False, stream=None):
        """d_windows: device int64 tensor [n, 2] (16-B records); d_out: device int32 tensor [n, 4]."""
        self._check(self._lib.smi_bc_match_device(self._h, _ptr(d_windows), int(n), int(max_ed),
                                                  int(bool(five_prime)), _ptr(d_out), _stream_ptr(stream)))

    def extract_windows_device(self, d_reads, d_offsets, d_adapter_end, d_windows, n, five_prime=False, stream=None):
        self._check(self._lib.smi_extract_windows_device(self._h, _ptr(d_reads), _ptr(d_offsets), _ptr(d_adapter_end),
                                                         int(n), int(bool(five_prime)), _ptr(d_windows),
                                                         _stream_ptr(stream)))

    def hist_device(self, d_keys_u32, d_pass_u8, d_hist_u32, n, stream=None):
        self._check(self._lib.smi_hist_device(self._h, _ptr(d_keys_u32), _ptr(d_pass_u8), int(n), _ptr(d_hist_u32),
                                              _stream_ptr(stream)))

    # ---- read scan ---------------------------------------------------------------------------------------
    def scan_config(self, pass_no=2, knobs=None, five_prime=False, dont_search_polya=False):
        """shipped config.xml values, or what the chunk workers derive from `knobs` (smi_scan_config_from_knobs); pass 1 = complete adapter
        (22 nt), pass 2 = short adapter (10 nt)"""
        cfg = np.zeros(1, dtype=SCAN_CONFIG_DTYPE)
        self._check(self._lib.smi_scan_config_from_knobs(ctypes.byref(knobs) if knobs is not None else None, int(pass_no), int(five_prime),
                                                         int(dont_search_polya), _ptr(cfg)))
        return cfg

    def pack_ends_device(self, d_reads, d_quals, d_offsets, n, d_ends, d_len, d_qtail=None, d_qsum=None, stream=None,
                         five_prime=False):
        self._check(self._lib.smi_pack_ends_device(self._h, _ptr(d_reads), _ptr(d_quals), _ptr(d_offsets), int(n),
                                                   int(five_prime), _ptr(d_ends), _ptr(d_len), _ptr(d_qtail), _ptr(d_qsum),
                                                   _stream_ptr(stream)))

    def scan_config_5p(self, pass_no=2, dont_search_polya=True):
        cfg = np.zeros(1, dtype=SCAN_CONFIG_DTYPE)
        self._check(self._lib.smi_scan_default_config_5p(int(pass_no), int(dont_search_polya), _ptr(cfg)))
        return cfg

    def scan_device(self, d_ends, d_len, n, cfg, d_out, d_windows=None, d_qtail=None, d_qsum=None, stream=None):
        """d_ends int32 [28, 2n]; d_len int32 [n]; d_out int32 [n, 8] (32-B records); d_windows int64 [n, 2]"""
        self._check(self._lib.smi_scan_device(self._h, _ptr(d_ends), _ptr(d_len), _ptr(d_qtail), _ptr(d_qsum), int(n),
                                              _ptr(cfg), _ptr(d_out), _ptr(d_windows), _stream_ptr(stream)))

    # ---- FASTQ ingest --------------------------------------------------------------------------------------
    def fastq_index_device(self, d_text, n_bytes, d_line_start, d_name_start, d_name_len, d_seq_start, d_seq_len, d_qual_start,
                           d_offsets, cap_records, stream=None):
        """-> (n_records, error bits); buffers: int64 / int32 device tensors of capacity cap_records (+1 for offsets)"""
        n_rec = ctypes.c_size_t(0)
        err = ctypes.c_uint32(0)
        self._check(self._lib.smi_fastq_index_device(self._h, _ptr(d_text), int(n_bytes), _ptr(d_line_start),
                                                     int(d_line_start.numel()), _ptr(d_name_start), _ptr(d_name_len),
                                                     _ptr(d_seq_start), _ptr(d_seq_len), _ptr(d_qual_start), _ptr(d_offsets),
                                                     int(cap_records), ctypes.byref(n_rec), ctypes.byref(err),
                                                     _stream_ptr(stream)))
        return n_rec.value, err.value

    def fastq_gather_device(self, d_text, d_start, d_offsets, n, d_out, stream=None):
        self._check(self._lib.smi_fastq_gather_device(self._h, _ptr(d_text), _ptr(d_start), _ptr(d_offsets), int(n), _ptr(d_out),
                                                      _stream_ptr(stream)))

    def pack_reads_text_device(self, d_text, d_seq_start, d_offsets, n, total_bases, d_planes, stream=None):
        """K-PACKR reading the bases where the FASTQ text has them (no gathered copy)"""
        self._check(self._lib.smi_pack_reads_text_device(self._h, _ptr(d_text), _ptr(d_seq_start), _ptr(d_offsets), int(n), int(total_bases),
                                                         _ptr(d_planes), _stream_ptr(stream)))

    def frag_text_starts_device(self, d_seq_start, d_qual_start, d_offsets, d_frag_offsets, d_frag_src, n_out, d_base_start, d_qual_out,
                                stream=None):
        self._check(self._lib.smi_frag_text_starts_device(self._h, _ptr(d_seq_start), _ptr(d_qual_start), _ptr(d_offsets), _ptr(d_frag_offsets),
                                                          _ptr(d_frag_src), int(n_out), _ptr(d_base_start), _ptr(d_qual_out),
                                                          _stream_ptr(stream)))

    def pack_ends_text_device(self, d_text, d_base_start, d_offsets, n, d_ends, d_len, stream=None):
        self._check(self._lib.smi_pack_ends_text_device(self._h, _ptr(d_text), _ptr(d_base_start), _ptr(d_offsets), int(n), _ptr(d_ends),
                                                        _ptr(d_len), _stream_ptr(stream)))

    def fastq_write_device(self, d_text, d_line_start, d_reads, d_quals, d_offsets, d_frag_src, d_chim, d_scan, d_bc, d_rank,
                           n_out, first_read_id, d_passed, d_failed, d_rec_off, d_is_passed, five_prime=False,
                           trim_fastq=False, stream=None, in_text=False):
        """K-WRITE -> (bytes passed, bytes failed, records passed); raises SmiError with the SMI_WR_* bits.
        in_text: d_reads / d_quals are the int64 text positions of frag_text_starts_device (smi_fastq_write_text_device)"""
        cfg = (ctypes.c_int32 * 2)(int(bool(five_prime)), int(bool(trim_fastq)))
        totals = (ctypes.c_uint64 * 3)()
        err = ctypes.c_uint32(0)
        opt = lambda t: _ptr(t) if t is not None else None  # noqa: E731
        fn = self._lib.smi_fastq_write_text_device if in_text else self._lib.smi_fastq_write_device
        rc = fn(self._h, _ptr(d_text), _ptr(d_line_start), _ptr(d_reads), _ptr(d_quals),
                                              _ptr(d_offsets), opt(d_frag_src), opt(d_chim), _ptr(d_scan), _ptr(d_bc),
                                              opt(d_rank), int(n_out), int(first_read_id), ctypes.byref(cfg), _ptr(d_passed),
                                              int(d_passed.numel()), _ptr(d_failed), int(d_failed.numel()), _ptr(d_rec_off),
                                              _ptr(d_is_passed), totals, ctypes.byref(err), _stream_ptr(stream))
        if rc != 0:
            raise SmiError(f"smi_fastq_write_device: {self._lib.smi_last_error().decode()} (error bits {err.value})")
        return int(totals[0]), int(totals[1]), int(totals[2])

    def bc_counts_device(self, d_results, n, d_counts, stream=None):
        """K-CNT: d_counts[3 * ordinal(bc) + ed] += 1 for the assigned reads of a batch"""
        self._check(self._lib.smi_bc_counts_device(self._h, _ptr(d_results), int(n), _ptr(d_counts), _stream_ptr(stream)))

    # ---- one native call per chunk (smi_worker.hip) ---------------------------------------------------------
    def scanfastq_pass2_chunk(self, text, max_ed=1, five_prime=False, dont_search_polya=False, split_chimeras=True, trim_fastq=False,
                              first_read_id=1, rank_keys=None, rank_values=None, want_results=False, copy=True, packed=False, n_threads=4, compress=False,
                              device_output=False):
        """host FASTQ bytes (or a numpy uint8 array, e.g. PinnedBuffer.array) -> (passed, failed, info dict); everything in
        between on the device.  copy=False returns numpy views of the context's pinned output buffers (valid until its next call).
        packed=True: smi_scanfastq_pass2_chunk_packed -- the host indexes / packs / writes on n_threads threads, the link carries
        bit-planes up and decisions down; same bytes out.  compress=True (text worker): `passed` / `failed` are one gzip member each (K-DEFLATE);
        info["passed_text_bytes"] / ["failed_text_bytes"] give the sizes of the text.  device_output=True (text worker): nothing is downloaded --
        passed / failed stay on the DEVICE (copy=True: uint8 tensors of their own; copy=False: DeviceSpan = address + size inside the context's
        arena, valid until its next call, .tensor() for a view); with a device tensor as `text` the chunk never touches the host"""
        cfg = Pass2Config()
        self._check(self._lib.smi_pass2_default_config(ctypes.byref(cfg)))
        cfg.max_ed, cfg.five_prime, cfg.dont_search_polya = int(max_ed), int(five_prime), int(dont_search_polya)
        cfg.split_chimeras, cfg.trim_fastq, cfg.want_results, cfg.first_read_id = int(split_chimeras), int(trim_fastq), int(want_results), int(first_read_id)
        cfg.compress = 1 if compress else 0
        cfg.device_output = 1 if device_output else 0
        keep = []
        if rank_keys is not None and len(rank_keys):
            k = np.ascontiguousarray(rank_keys, dtype=np.uint64)
            v = np.ascontiguousarray(rank_values, dtype=np.int32)
            keep += [k, v]
            cfg.rank_keys, cfg.rank_values, cfg.n_ranks = k.ctypes.data, v.ctypes.data, k.size
        out = Pass2Output()
        if hasattr(text, "is_cuda"):   # a uint8 device tensor (K-INFLATE's output): text worker only
            if packed or not text.is_cuda:
                raise SmiError("scanfastq_pass2_chunk: a tensor argument must be a device tensor, and only the text worker takes one")
            self._check(self._lib.smi_scanfastq_pass2_chunk(self._h, text.data_ptr(), int(text.numel()), ctypes.byref(cfg), ctypes.byref(out)))
        else:
            buf = text if isinstance(text, np.ndarray) else np.frombuffer(text, dtype=np.uint8)
            if packed:
                self._check(self._lib.smi_scanfastq_pass2_chunk_packed(self._h, buf.ctypes.data, buf.size, ctypes.byref(cfg), int(n_threads),
                                                                       ctypes.byref(out)))
            else:
                self._check(self._lib.smi_scanfastq_pass2_chunk(self._h, buf.ctypes.data, buf.size, ctypes.byref(cfg), ctypes.byref(out)))
        if device_output:
            passed, failed = (self._device_bytes(out.passed, out.passed_bytes, copy), self._device_bytes(out.failed, out.failed_bytes, copy))
        elif copy:
            passed = ctypes.string_at(out.passed, out.passed_bytes) if out.passed_bytes else b""
            failed = ctypes.string_at(out.failed, out.failed_bytes) if out.failed_bytes else b""
        else:
            view = lambda p, n: np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), shape=(n,)) if n else np.zeros(0, np.uint8)  # noqa: E731
            passed, failed = view(out.passed, out.passed_bytes), view(out.failed, out.failed_bytes)
        info = dict(n_records_in=out.n_records_in, n_records_out=out.n_records_out, n_passed=out.n_passed, passed_text_bytes=out.passed_text_bytes,
                    failed_text_bytes=out.failed_text_bytes)
        if want_results and out.stats:
            info["stats"] = np.frombuffer(ctypes.string_at(out.stats, 8 * N_SCAN_STATS), dtype=np.uint64).copy()
        if want_results and out.n_records_out:
            info["scan"] = np.frombuffer(ctypes.string_at(out.scan, out.n_records_out * SCAN_RESULT_DTYPE.itemsize), dtype=SCAN_RESULT_DTYPE)
            info["bc"] = np.frombuffer(ctypes.string_at(out.bc, out.n_records_out * BC_RESULT_DTYPE.itemsize), dtype=BC_RESULT_DTYPE)
        return passed, failed, info

    def _device_bytes(self, ptr, n, copy):
        """n bytes of the context's device memory: a DeviceSpan (address + size, valid until the context's next call; .tensor() is a torch view) or,
        with copy, a uint8 tensor of their own"""
        span = DeviceSpan(int(ptr or 0), int(n), self.device)
        if not copy:
            return span
        import torch

        c = span.tensor().clone()   # the worker's stream had drained when it returned; the copy is on torch's stream and must be over before the arena is reused
        torch.cuda.current_stream(c.device).synchronize()
        return c

    def scanfastq_pass1_chunk(self, text, d_hist, five_prime=False, dont_search_polya=False, packed=False, n_threads=4):
        """adds the chunk's whitelist hits to d_hist (int32 device tensor, one counter per loaded key) -> n records"""
        n, err = ctypes.c_size_t(0), ctypes.c_uint32(0)
        if hasattr(d_hist, "is_cuda") and d_hist.is_cuda:  # the zero-fill of d_hist ran on torch's stream (sicelore_mi.h)
            import torch

            torch.cuda.current_stream(d_hist.device).synchronize()
        if hasattr(text, "is_cuda"):   # a uint8 device tensor (K-INFLATE's output): text worker only
            if packed or not text.is_cuda:
                raise SmiError("scanfastq_pass1_chunk: a tensor argument must be a device tensor, and only the text worker takes one")
            self._check(self._lib.smi_scanfastq_pass1_chunk(self._h, text.data_ptr(), int(text.numel()), int(five_prime), int(dont_search_polya),
                                                            _ptr(d_hist), ctypes.byref(n), ctypes.byref(err)))
            return n.value
        buf = _as_u8(text)
        if packed:
            self._check(self._lib.smi_scanfastq_pass1_chunk_packed(self._h, buf.ctypes.data, buf.size, int(five_prime), int(dont_search_polya),
                                                                   _ptr(d_hist), int(n_threads), ctypes.byref(n), ctypes.byref(err)))
        else:
            self._check(self._lib.smi_scanfastq_pass1_chunk(self._h, buf.ctypes.data, buf.size, int(five_prime), int(dont_search_polya),
                                                            _ptr(d_hist), ctypes.byref(n), ctypes.byref(err)))
        return n.value

    def scanfastq_pass1_chunk_keys(self, text, d_keys, d_count, five_prime=False, dont_search_polya=False):
        """pass 1 of a chunk WITHOUT a list of possible barcodes (`-a none`): the barcode of every read that passes the filter is appended to
        d_keys (int64 device tensor; the reference's long) and counted in d_count (int64 device tensor of one element, zeroed by the caller and
        shared by the chunks of a pass) -> n records.  text: host bytes / numpy uint8 or a uint8 device tensor"""
        import torch

        n, err = ctypes.c_size_t(0), ctypes.c_uint32(0)
        torch.cuda.current_stream(d_keys.device).synchronize()      # (d_count was zeroed on torch's stream: sicelore_mi.h, smi_scanfastq_pass1_chunk)
        if hasattr(text, "is_cuda"):
            ptr, nb = text.data_ptr(), int(text.numel())
        else:
            buf = _as_u8(text)
            ptr, nb = buf.ctypes.data, buf.size
        self._check(self._lib.smi_scanfastq_pass1_chunk_keys(self._h, ptr, nb, int(five_prime), int(dont_search_polya), _ptr(d_keys), int(d_keys.numel()),
                                                             _ptr(d_count), ctypes.byref(n), ctypes.byref(err)))
        return n.value

    def pass1_keys_device(self, d_windows, d_scan, n, d_keys, d_count, stream=None):
        self._check(self._lib.smi_pass1_keys_device(self._h, _ptr(d_windows), _ptr(d_scan), int(n), _ptr(d_keys), int(d_keys.numel()), _ptr(d_count),
                                                    _stream_ptr(stream)))

    def count_keys_device(self, d_keys, n):
        """the key list of a pass -> (distinct keys ascending, their counts) as numpy uint64 / uint32 (smi_count_keys_device: radix sort + run lengths)"""
        import torch

        n = int(n)
        if n == 0:
            return np.zeros(0, dtype=np.uint64), np.zeros(0, dtype=np.uint32)
        uniq = torch.empty(n, dtype=torch.int64, device=d_keys.device)
        cnt = torch.empty(n, dtype=torch.int32, device=d_keys.device)
        nu = torch.zeros(1, dtype=torch.int64, device=d_keys.device)
        torch.cuda.current_stream(d_keys.device).synchronize()
        self._check(self._lib.smi_count_keys_device(self._h, _ptr(d_keys), n, _ptr(uniq), _ptr(cnt), _ptr(nu), None))
        m = int(nu.item())
        return uniq[:m].cpu().numpy().view(np.uint64), cnt[:m].cpu().numpy().view(np.uint32)

    def ends_from_planes_device(self, d_planes, d_read_offsets, n_reads, total_bases, d_rec_offsets, d_frag_src, n_records, d_ends, d_len,
                                stream=None):
        """K-PACK from read planes (smi_ends_from_planes_device)"""
        self._check(self._lib.smi_ends_from_planes_device(self._h, _ptr(d_planes), _ptr(d_read_offsets), int(n_reads), int(total_bases),
                                                          _ptr(d_rec_offsets), _ptr(d_frag_src), int(n_records), _ptr(d_ends), _ptr(d_len),
                                                          _stream_ptr(stream)))

    def bgzf_deflate_device(self, data, out=None):
        """smi_bgzf_deflate_device: bytes / uint8 array -> the BGZF stream (numpy uint8), blocks deflated on the device.  out: a uint8 array to
        write into (e.g. page-locked memory that is reused from call to call; a view of it is returned) -- it must hold bgzf_device_bound(n)"""
        a = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, dtype=np.uint8)
        n = ctypes.c_size_t(0)
        self._check(self._lib.smi_bgzf_deflate_device(self._h, a.ctypes.data if a.size else None, a.size, None, 0, ctypes.byref(n)))
        if out is None or out.size < n.value:
            out = np.empty(n.value, dtype=np.uint8)
        self._check(self._lib.smi_bgzf_deflate_device(self._h, a.ctypes.data if a.size else None, a.size, out.ctypes.data, out.size, ctypes.byref(n)))
        return out[:n.value]

    @staticmethod
    def bgzf_device_bound(n_bytes):
        """bytes smi_bgzf_deflate_device may write for n_bytes of input (its own bound: blocks of 61,440 input bytes)"""
        n = int(n_bytes)
        return n + n // 8 + ((n + 0xF000 - 1) // 0xF000) * 628 + 92

    def gz_inflate_device(self, files=None, out_caps=None, packed=None):
        """K-INFLATE (smi_gz_inflate_device): a list of gzip files (bytes / uint8 arrays) -> (uint8 device tensor with all texts, offsets, lengths,
        statuses); file i's text is out[offsets[i] : offsets[i] + lengths[i]] when statuses[i] == 0.  out_caps: capacity per file (default: the
        ISIZE field at the end of the file, which is the text's size for a single-member file below 4 GB).  packed: what pack_gz_paths made of
        the files (read and laid out by another thread while the device was busy), instead of `files`"""
        import torch

        if packed is None:
            arrs = [np.frombuffer(f, dtype=np.uint8) if not isinstance(f, np.ndarray) else f for f in files]
            in_off, at = [], 0
            for a in arrs:
                in_off.append(at)
                at = (at + a.size + 511) & ~511
            host = np.zeros(at + 1024, dtype=np.uint8)
            for a, o in zip(arrs, in_off):
                host[o:o + a.size] = a
            sizes = [a.size for a in arrs]
            if out_caps is None:
                out_caps = [int.from_bytes(a[-4:].tobytes(), "little") if a.size >= 18 else 0 for a in arrs]
                out_caps = [c if c <= 1032 * a.size else 0 for c, a in zip(out_caps, arrs)]   # an untrusted field: never more than DEFLATE can expand to
        else:
            host, in_off, sizes = packed["host"], packed["in_off"], packed["sizes"]
            out_caps = packed["out_caps"] if out_caps is None else out_caps
        n = len(sizes)
        out_off, at = [], 0
        for c in out_caps:
            out_off.append(at)
            at = (at + int(c) + 255) & ~255
        dev = torch.device("cuda", self.device)
        d_in = torch.from_numpy(host).to(dev)
        d_out = torch.empty(max(at, 1), dtype=torch.uint8, device=dev)
        S = np.zeros((n, 4), dtype=np.uint64)
        S[:, 0], S[:, 1], S[:, 2], S[:, 3] = in_off, sizes, out_off, out_caps
        R = np.zeros(n, dtype=np.dtype([("out_len", "<u8"), ("status", "<u4"), ("n_members", "<u4")]))
        torch.cuda.current_stream().synchronize()      # the upload (on the caller's torch stream) before the context's own stream reads it
        self._check(self._lib.smi_gz_inflate_device(self._h, _ptr(d_in), S.ctypes.data, n, _ptr(d_out), R.ctypes.data, None))
        return d_out, np.array(out_off, dtype=np.int64), R["out_len"].astype(np.int64), R["status"].copy(), R["n_members"].copy()

    def gzip_device(self, d_in, n_bytes=None, raw_deflate=False, d_out=None, stream=None):
        """K-DEFLATE (smi_gzip_device): a uint8 device tensor -> one gzip member (or raw deflate stream) as a uint8 device tensor view"""
        import torch

        n = int(d_in.numel() if n_bytes is None else n_bytes)
        cap = int(self._lib.smi_deflate_bound(n))
        if d_out is None:
            d_out = torch.empty(cap, dtype=torch.uint8, device=d_in.device)
        total = torch.zeros(2, dtype=torch.int64, device=d_in.device)
        self._check(self._lib.smi_gzip_device(self._h, _ptr(d_in), n, _ptr(d_out), int(d_out.numel()), _ptr(total), 1 if raw_deflate else 0,
                                              _stream_ptr(stream)))
        t = total.cpu()
        if int(t[1]):
            raise SmiError(f"smi_gzip_device: error flags {int(t[1])}")
        return d_out[:int(t[0])]

    def assignumis_chunk_raw(self, inp, keep_data_end=False, max_dist=500, bc_edit_limit=None, n_threads=4, five_prime=False, cluster_cfg=None,
                             umi_length=0, grouping_distance=None, random_umi_seed=0):
        """the same on the buffers of bam_chunk_inputs -> (UMI_TAG_DTYPE array, n_done)"""
        n = inp["n"]
        cfg = AssignUmisConfig()
        self._check(self._lib.smi_assignumis_default_config(ctypes.byref(cfg)))
        cfg.max_dist, cfg.keep_data_end, cfg.n_threads = int(max_dist), int(keep_data_end), int(n_threads)
        cfg.five_prime = int(bool(five_prime))
        cfg.bc_edit_limit = -1 if bc_edit_limit is None else int(bc_edit_limit)
        cfg.umi_length = int(umi_length)               # umis/umi_length (0: the context's knob, 12 without one)
        cfg.random_umi_seed = int(random_umi_seed)     # assignumis -f (0: off)
        if grouping_distance is not None:              # barcodes/distance_from_read_end_for_grouping
            cfg.grouping_distance = int(grouping_distance)
        if cluster_cfg is not None:
            cluster_cfg = np.ascontiguousarray(cluster_cfg, dtype=UMI_CLUSTER_CONFIG_DTYPE)
            cfg.cluster = cluster_cfg.ctypes.data
        out = np.zeros(max(n, 1), dtype=UMI_TAG_DTYPE)
        nd = ctypes.c_int32(0)
        self._check(self._lib.smi_assignumis_chunk(self._h, inp["names"].ctypes.data, inp["name_off"].ctypes.data, inp["flags"].ctypes.data,
                                                   inp["pos0"].ctypes.data, inp["cigars"].ctypes.data, inp["cigar_off"].ctypes.data, n,
                                                   ctypes.byref(cfg), out.ctypes.data, ctypes.byref(nd)))
        return out[:n], nd.value

    def assignumis_chunk(self, names, flags, pos0, cigars, keep_data_end=False, max_dist=500, bc_edit_limit=None, n_threads=4,
                         five_prime=False, cluster_cfg=None, umi_length=0, grouping_distance=None, random_umi_seed=0):
        """one BamReader chunk through the native worker -> (UMI_TAG_DTYPE array, n_done); names: list of QNAME strings,
        cigars: list of numpy uint32 arrays (BAM encoding)"""
        n = len(names)
        enc = [nm.encode() for nm in names]
        noff = np.zeros(n + 1, dtype=np.uint32)
        noff[1:] = np.cumsum([len(e) for e in enc])
        nbuf = np.frombuffer(b"".join(enc) + b"\0", dtype=np.uint8)
        coff = np.zeros(n + 1, dtype=np.uint32)
        coff[1:] = np.cumsum([len(c) for c in cigars])
        cbuf = np.ascontiguousarray(np.concatenate([np.asarray(c, dtype=np.uint32) for c in cigars] + [np.zeros(1, np.uint32)]))
        fl = np.ascontiguousarray(flags, dtype=np.uint16)
        p0 = np.ascontiguousarray(pos0, dtype=np.int32)
        cfg = AssignUmisConfig()
        self._check(self._lib.smi_assignumis_default_config(ctypes.byref(cfg)))
        cfg.max_dist, cfg.keep_data_end, cfg.n_threads = int(max_dist), int(keep_data_end), int(n_threads)
        cfg.five_prime = int(bool(five_prime))
        cfg.bc_edit_limit = -1 if bc_edit_limit is None else int(bc_edit_limit)
        cfg.umi_length = int(umi_length)               # umis/umi_length (0: the context's knob, 12 without one)
        cfg.random_umi_seed = int(random_umi_seed)     # assignumis -f (0: off)
        if grouping_distance is not None:              # barcodes/distance_from_read_end_for_grouping
            cfg.grouping_distance = int(grouping_distance)
        if cluster_cfg is not None:  # umi_cluster_config(...) record: the clusterer's knobs (shipped values otherwise)
            cluster_cfg = np.ascontiguousarray(cluster_cfg, dtype=UMI_CLUSTER_CONFIG_DTYPE)
            cfg.cluster = cluster_cfg.ctypes.data
        out = np.zeros(max(n, 1), dtype=UMI_TAG_DTYPE)
        nd = ctypes.c_int32(0)
        self._check(self._lib.smi_assignumis_chunk(self._h, nbuf.ctypes.data, noff.ctypes.data, fl.ctypes.data, p0.ctypes.data,
                                                   cbuf.ctypes.data, coff.ctypes.data, n, ctypes.byref(cfg), out.ctypes.data,
                                                   ctypes.byref(nd)))
        return out[:n], nd.value

    # ---- chimera splitter ----------------------------------------------------------------------------------
    def chimera_config(self, five_prime=False, knobs=None):
        """the splitter's configuration: shipped, or from `knobs` (its strings then point into that object: it is kept alive on the result)"""
        cfg = ChimeraConfig()
        self._check(self._lib.smi_chimera_config_from_knobs(ctypes.byref(knobs) if knobs is not None else None, int(five_prime), ctypes.byref(cfg)))
        cfg._knobs = knobs
        return cfg

    def read_planes_words(self, total_bases, n):
        return int(self._lib.smi_read_planes_words(int(total_bases), int(n)))

    def pack_reads_device(self, d_reads, d_offsets, n, total_bases, d_planes, stream=None):
        """d_planes: int32 [read_planes_words(total_bases, n)]"""
        self._check(self._lib.smi_pack_reads_device(self._h, _ptr(d_reads), _ptr(d_offsets), int(n), int(total_bases),
                                                    _ptr(d_planes), _stream_ptr(stream)))

    def chimera_device(self, d_planes, d_offsets, n, total_bases, cfg, d_out, stream=None):
        """d_out: int32 [n, 4] (16-B smi_chimera_result records)"""
        self._check(self._lib.smi_chimera_device(self._h, _ptr(d_planes), _ptr(d_offsets), int(n), int(total_bases),
                                                 ctypes.addressof(cfg), _ptr(d_out), _stream_ptr(stream)))

    def split_offsets_device(self, d_chim, d_offsets, n, d_scratch, d_n_frag, d_frag_offsets, d_frag_src=None, stream=None):
        self._check(self._lib.smi_split_offsets_device(self._h, _ptr(d_chim), _ptr(d_offsets), int(n), _ptr(d_scratch),
                                                       _ptr(d_n_frag), _ptr(d_frag_offsets), _ptr(d_frag_src),
                                                       _stream_ptr(stream)))

    def hist_windows_device(self, d_windows, d_scan, n, d_hist, stream=None):
        self._check(self._lib.smi_hist_windows_device(self._h, _ptr(d_windows), _ptr(d_scan), int(n), _ptr(d_hist),
                                                      _stream_ptr(stream)))

    # ---- UMI pair distances ------------------------------------------------------------------------------
    @staticmethod
    def umi_offsets(group_sizes, padded=False):
        """host helper: group sizes -> (group_off uint32, pair_off uint64, mat_off uint64) prefix arrays; padded: the layout of
        smi_umi_dist_device_padded (rows of a group above 64 reads rounded up to 64 bytes, every group on a 64-byte boundary)"""
        n = np.asarray(group_sizes, dtype=np.uint64)
        go = np.zeros(n.size + 1, dtype=np.uint32)
        go[1:] = np.cumsum(n)
        po = np.zeros(n.size + 1, dtype=np.uint64)
        po[1:] = np.cumsum(n * (n + 1) // 2)
        mo = np.zeros(n.size + 1, dtype=np.uint64)
        if padded:
            ld = np.where(n > 64, (n + np.uint64(63)) & ~np.uint64(63), n)
            mo[1:] = np.cumsum((ld * n + np.uint64(63)) & ~np.uint64(63))
        else:
            mo[1:] = np.cumsum(n * n)
        return go, po, mo

    def umi_dist_device(self, d_windows, d_group_off, d_pair_off, d_mat_off, n_groups, total_pairs, d_out, stream=None, padded=False):
        fn = self._lib.smi_umi_dist_device_padded if padded else self._lib.smi_umi_dist_device
        self._check(fn(self._h, _ptr(d_windows), _ptr(d_group_off), _ptr(d_pair_off), _ptr(d_mat_off), int(n_groups), int(total_pairs), _ptr(d_out),
                       _stream_ptr(stream)))

    def umi_cluster_groups_device(self, d_dist, d_mat_off, d_group_off, n_groups, d_qv, d_out, d_skipped, cfg=None, stream=None):
        """K-UCLUST: d_out int64-viewable [n_reads] of 8-byte smi_umi_assignment records, d_skipped uint8 [n_reads]"""
        cfg = umi_cluster_config() if cfg is None else cfg
        self._check(self._lib.smi_umi_cluster_groups_device(self._h, _ptr(d_dist), _ptr(d_mat_off), _ptr(d_group_off), int(n_groups), _ptr(d_qv),
                                                            _ptr(cfg), _ptr(d_out), _ptr(d_skipped), _stream_ptr(stream)))

    # ---- host-buffer forms ---------------------------------------------------------------------------------
    def scan_batch(self, bases, quals, offsets, cfg, want_windows=True):
        """smi_scan_batch: numpy uint8 bases (+ qualities or None), uint64 offsets [n+1] -> (SCAN_RESULT_DTYPE [n], BC_WINDOW_DTYPE [n] or None)"""
        b = np.ascontiguousarray(bases, dtype=np.uint8)
        q = None if quals is None else np.ascontiguousarray(quals, dtype=np.uint8)
        o = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = o.size - 1
        out = np.zeros(max(n, 1), dtype=SCAN_RESULT_DTYPE)
        win = np.zeros(max(n, 1), dtype=BC_WINDOW_DTYPE) if want_windows else None
        self._check(self._lib.smi_scan_batch(self._h, _ptr(b), _ptr(q), _ptr(o), n, _ptr(cfg), _ptr(out), _ptr(win)))
        return out[:n], (None if win is None else win[:n])

    def umi_dist_batch(self, windows, group_off):
        """smi_umi_dist_batch: packed windows (uint64) of all groups, group_off uint32 [n_groups + 1] -> the groups' matrices back to back"""
        w = np.ascontiguousarray(windows, dtype=np.uint64)
        go = np.ascontiguousarray(group_off, dtype=np.uint32)
        sizes = np.diff(go.astype(np.int64))
        out = np.zeros(max(int((sizes * sizes).sum()), 1), dtype=np.uint8)
        self._check(self._lib.smi_umi_dist_batch(self._h, _ptr(w), _ptr(go), go.size - 1, _ptr(out)))
        return out[:int((sizes * sizes).sum())]

    # ---- timing ------------------------------------------------------------------------------------------
    def set_timing(self, enabled=True):
        self._check(self._lib.smi_set_timing(self._h, int(bool(enabled))))

    K_BC_MATCH, K_SCAN, K_HIST, K_PACK, K_UMI, K_CHIMERA = 0, 1, 2, 3, 4, 5

    def kernel_ms(self, kernel_id):
        ms = ctypes.c_float(-1.0)
        self._check(self._lib.smi_kernel_ms(self._h, int(kernel_id), ctypes.byref(ms)))
        return float(ms.value)

    def last_kernel_ms(self):
        ms = ctypes.c_float(-1.0)
        self._check(self._lib.smi_last_kernel_ms(self._h, ctypes.byref(ms)))
        return float(ms.value)


def hist_allreduce(contexts, d_hists):
    """smi_hist_allreduce_after: in-place sum of the pass-1 histograms (device int32/uint32 tensors, one per context / GPU) over RCCL;
    every context's stream is ordered behind torch's current stream on the tensor's device (the stream that filled it)"""
    import torch

    lib = load_library()
    n = len(contexts)
    hs = (ctypes.c_void_p * n)(*[c._h for c in contexts])
    ps = (ctypes.c_void_p * n)(*[ctypes.c_void_p(t.data_ptr()) for t in d_hists])
    ss = (ctypes.c_void_p * n)(*[ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream) for t in d_hists])
    rc = lib.smi_hist_allreduce_after(hs, n, ps, int(d_hists[0].numel()), ss)
    if rc != 0:
        raise SmiError(f"smi_hist_allreduce error {rc}: {lib.smi_last_error().decode()}")


def hist_allreduce_release():
    """destroy the cached RCCL communicators (smi_hist_allreduce_release)"""
    load_library().smi_hist_allreduce_release()
