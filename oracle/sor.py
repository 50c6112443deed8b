"""ctypes binding of the ORACLE (oracle/liboracle.so).  Test infrastructure: import only from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")

MATCH_DTYPE = np.dtype([("read_seq", "<i8"), ("matching_bc", "<i8"), ("ed", "<i4"), ("subs", "<i4"), ("ins", "<i4"),
                        ("dels", "<i4"), ("offset", "<i4"), ("length", "<i4")])
ASSIGN_DTYPE = np.dtype([("bc", "<i8"), ("found", "<i4"), ("ed", "<i4"), ("ed_sec", "<i4"), ("offset", "<i4"),
                         ("ins_minus_del", "<i4"), ("bc_start", "<i4"), ("bc_end", "<i4"), ("n_matches", "<i4"),
                         ("n_probes", "<u8")])
assert MATCH_DTYPE.itemsize == 40 and ASSIGN_DTYPE.itemsize == 48


def build(force=False):
    if force or not os.path.exists(LIB_PATH) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(LIB_PATH)
        for f in os.listdir(_HERE) if f.endswith((".c", ".h"))
    ):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = ctypes.CDLL(LIB_PATH)
        i64, ci, vp, sz = ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t
        L.sor_twobit_encode.restype = i64
        L.sor_twobit_encode.argtypes = [ctypes.c_char_p, ci]
        L.sor_twobit_decode.argtypes = [i64, ci, ctypes.c_char_p]
        L.sor_twobit_revcomp.restype = i64
        L.sor_twobit_revcomp.argtypes = [i64, ci]
        L.sor_replace_deg.argtypes = [i64, ci, ci, vp]
        L.sor_insert_deg.argtypes = [i64, ci, ci, vp]
        L.sor_delete_byte.restype = i64
        L.sor_delete_byte.argtypes = [i64, ci, ci, ci]
        L.sor_set_new.restype = vp
        L.sor_set_new.argtypes = [vp, sz]
        L.sor_set_free.argtypes = [vp]
        L.sor_set_contains.argtypes = [vp, i64]
        L.sor_bc_match.argtypes = [vp, i64, ci, ci, ci, ci, vp, ci, ci, ci, vp, ci, vp]
        L.sor_assign_barcode.argtypes = [vp, ctypes.c_char_p, ci, ci, ci, ci, ci, ci, vp]
        L.sor_assign_batch_codes.argtypes = [vp, vp, ci, vp, sz, ci, ci, ci, vp, vp, ci]
        _LIB = L
    return _LIB


def encode(s):
    return int(lib().sor_twobit_encode(s.encode(), len(s)))


def decode(v, n=16):
    b = ctypes.create_string_buffer(n + 1)
    lib().sor_twobit_decode(ctypes.c_int64(v), n, b)
    return b.value.decode()


def revcomp(v, n=16):
    return int(lib().sor_twobit_revcomp(ctypes.c_int64(v), n))


def replace_deg(seq, pos, length=16):
    out = np.zeros(4, dtype=np.int64)
    lib().sor_replace_deg(ctypes.c_int64(seq), pos, length, out.ctypes.data)
    return out


def insert_deg(seq, pos, length=16):
    out = np.zeros(4, dtype=np.int64)
    lib().sor_insert_deg(ctypes.c_int64(seq), pos, length, out.ctypes.data)
    return out


def delete_byte(seq, base4, pos, length=16):
    return int(lib().sor_delete_byte(ctypes.c_int64(seq), base4, pos, length))


class BarcodeSet:
    def __init__(self, keys):
        k = np.ascontiguousarray(np.asarray(keys, dtype=np.int64))
        self._k = k
        self._h = lib().sor_set_new(k.ctypes.data, k.size)

    def __contains__(self, key):
        return bool(lib().sor_set_contains(self._h, ctypes.c_int64(int(key))))

    def __del__(self):
        try:
            lib().sor_set_free(self._h)
        except Exception:
            pass


def bc_match(bset, seq, ed, post4=None, offset=0, skip_full=False, allow_indels=True, do_next=True, length=16):
    """BarcodeMatchTester.call -> (matches in HashSet iteration order, number of set probes)"""
    out = np.zeros(64, dtype=MATCH_DTYPE)
    np_ = ctypes.c_uint64(0)
    if post4 is None:
        pp, pl = None, 0
    else:
        post = np.ascontiguousarray(np.asarray(post4, dtype=np.uint8))
        pp, pl = post.ctypes.data, post.size
    n = lib().sor_bc_match(bset._h, ctypes.c_int64(int(seq)), length, ed, int(skip_full), int(allow_indels), pp, pl,
                           offset, int(do_next), out.ctypes.data, 64, ctypes.byref(np_))
    if n < 0:
        raise RuntimeError(f"sor_bc_match: {n}")
    return out[:n].copy(), int(np_.value)


def assign_barcode(bset, stranded, adapterpos, max_ed=1, test_pm=2, five_prime=False):
    """Parser.assignBarcode -> (status, record); status -1 = the reference would throw"""
    r = np.zeros(1, dtype=ASSIGN_DTYPE)
    s = stranded.encode() if isinstance(stranded, str) else stranded
    rc = lib().sor_assign_barcode(bset._h, s, len(s), adapterpos, max_ed, test_pm, int(five_prime), 16, r.ctypes.data)
    return rc, r[0]


def assign_batch(bset, codes, ae, max_ed=1, test_pm=2, five_prime=False, n_threads=1):
    """codes: uint8 [n, width]; ae: int32 [n] -> (status int32 [n], records ASSIGN_DTYPE [n])"""
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    ae = np.ascontiguousarray(ae, dtype=np.int32)
    n, width = codes.shape
    out = np.zeros(n, dtype=ASSIGN_DTYPE)
    st = np.zeros(n, dtype=np.int32)
    rc = lib().sor_assign_batch_codes(bset._h, codes.ctypes.data, width, ae.ctypes.data, n, max_ed, test_pm,
                                      int(five_prime), out.ctypes.data, st.ctypes.data, n_threads)
    if rc != 0:
        raise RuntimeError("sor_assign_batch_codes failed")
    return st, out


# ---- read scan (sor_scan.c) ----------------------------------------------------------------------------------
SCAN_PARAMS_DTYPE = np.dtype([("min_read_length", "<i4"), ("polya_len", "<i4"), ("polya_frac", "<f4"),
                              ("window_polya", "<i4"), ("min_adapter_3p_matches", "<i4"), ("min_mean_bc_qv", "<i4"),
                              ("min_mean_read_qv", "<i4"), ("tso", "S20"), ("tso_window", "<i4"), ("tso_max_mm", "<i4"),
                              ("tso_min_consec", "<i4"), ("tso_min_two", "<i4")])
SCAN_RESULT_DTYPE = np.dtype([("flags", "<u8"), ("adapter_found", "<i4"), ("reverse", "<i4"), ("polya_start", "<i4"),
                              ("polya_end", "<i4"), ("adapter_start", "<i4"), ("adapter_end", "<i4"),
                              ("scan_end", "<i4"), ("adapter_nmis", "<i4"), ("n_cand_fwd", "<i4"),
                              ("n_cand_rev", "<i4"), ("pass1_ok", "<i4"), ("mean_qv_bc", "<f4"),
                              ("mean_qv_read", "<f4"), ("tso_start", "<i4"), ("tso_end", "<i4")], align=True)
assert SCAN_RESULT_DTYPE.itemsize == 72
FLAG_BITS = {"FAILED": 5, "PASSED_FWD": 8, "PASSED_REV": 9, "POLY_T_5P": 11, "POLY_A_3P": 12, "POLY_A_NOT_FOUND": 13,
             "POLY_T_5P_POLY_A_3P": 14, "ADAPTER_5P": 15, "ADAPTER_3P": 16, "ADAPTER_SELECTED_DESP_BOTH": 19,
             "READ_TOO_SHORT": 20, "ADAPTER_5P_AND_3P": 21, "TSO_5P": 17, "TSO_3P": 18, "TSO_5P_AND_3P": 22}


def default_scan_params():
    p = np.zeros(1, dtype=SCAN_PARAMS_DTYPE)
    p[0] = (200, 15, 0.75, 150, 8, 8, 8, b"", 0, 0, 0, 0)  # Jar/config.xml:21,95-105,55-59; an empty `tso` = the shipped TSO parameters (:155-166)
    return p


def set_tso_params(p, sequence="AACGCAGAGTACATGG", window=90, max_mm=5, min_consec=8, min_two=12):
    """the read scan's TSO parameters (tso_for3pBarcoding) on a scan-parameter record"""
    p["tso"] = sequence.encode()
    p["tso_window"], p["tso_max_mm"], p["tso_min_consec"], p["tso_min_two"] = window, max_mm, min_consec, min_two
    return p


def _scan_sigs():
    L = lib()
    vp, ci, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
    L.sor_scan_read_3p.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ci, ctypes.c_char_p, ci, vp, vp]
    L.sor_scan_batch_3p.argtypes = [vp, vp, vp, sz, ctypes.c_char_p, ci, vp, vp, vp, ci]
    L.sor_find_polyt.argtypes = [vp, ci, ci, ctypes.c_float, ci, ctypes.POINTER(ci), ctypes.POINTER(ci)]
    L.sor_nw_strings.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p,
                                 ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ci), ctypes.POINTER(ci),
                                 ctypes.POINTER(ci), ctypes.POINTER(ctypes.c_float)]
    return L


def scan_read_3p(read, qual, adapter, max_mm=3, params=None):
    L = _scan_sigs()
    p = default_scan_params() if params is None else params
    r = np.zeros(1, dtype=SCAN_RESULT_DTYPE)
    rc = L.sor_scan_read_3p(read.encode(), qual.encode() if qual is not None else None, len(read), adapter.encode(),
                            max_mm, p.ctypes.data, r.ctypes.data)
    return rc, r[0]


def scan_read_5p(read, qual, adapter, max_mm=4, params=None, window=110, dont_search_polya=True):
    """5' barcoding; max_mm = maxNeedlemanMismatches + 1 (Parser.java:L99)"""
    L = _scan_sigs()
    L.sor_scan_read_5p.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int,
                                   ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    p = default_scan_params() if params is None else params
    r = np.zeros(1, dtype=SCAN_RESULT_DTYPE)
    rc = L.sor_scan_read_5p(read.encode(), qual.encode() if qual is not None else None, len(read), adapter.encode(),
                            max_mm, p.ctypes.data, window, int(dont_search_polya), r.ctypes.data)
    return rc, r[0]


def scan_batch_3p(reads_ascii, quals_ascii, offsets, adapter, max_mm=3, params=None, n_threads=1):
    """reads_ascii / quals_ascii: uint8 arrays (concatenated), offsets: uint64 [n+1]"""
    L = _scan_sigs()
    p = default_scan_params() if params is None else params
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    n = offsets.size - 1
    out = np.zeros(n, dtype=SCAN_RESULT_DTYPE)
    st = np.zeros(n, dtype=np.int32)
    reads_ascii = np.ascontiguousarray(reads_ascii, dtype=np.uint8)
    q = None if quals_ascii is None else np.ascontiguousarray(quals_ascii, dtype=np.uint8)
    L.sor_scan_batch_3p(reads_ascii.ctypes.data, None if q is None else q.ctypes.data, offsets.ctypes.data, n,
                        adapter.encode(), max_mm, p.ctypes.data, out.ctypes.data, st.ctypes.data, n_threads)
    return st, out


def find_polyt(codes4, minlen=15, minfrac=0.75, window=150):
    L = _scan_sigs()
    a = np.ascontiguousarray(codes4, dtype=np.uint8)
    b, e = ctypes.c_int(0), ctypes.c_int(0)
    rc = L.sor_find_polyt(a.ctypes.data, a.size, minlen, minfrac, window, ctypes.byref(b), ctypes.byref(e))
    return (b.value, e.value) if rc == 1 else None


def nw_strings(adapter, read_slice):
    L = _scan_sigs()
    a1, d, a2 = (ctypes.create_string_buffer(200) for _ in range(3))
    ne, e5 = ctypes.c_float(0), ctypes.c_float(0)
    i, dl, s = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    L.sor_nw_strings(adapter.encode(), read_slice.encode(), a1, d, a2, ctypes.byref(ne), ctypes.byref(i),
                     ctypes.byref(dl), ctypes.byref(s), ctypes.byref(e5))
    return a1.value.decode(), d.value.decode(), a2.value.decode(), ne.value, i.value, dl.value, s.value, e5.value


# ---- pass-1 finalize (sor_final.c) ---------------------------------------------------------------------------
def finalize_used_list(keys, counts, record_count, merge_ed=1, min_count_fold=10, cells_fold_below_max=500):
    L = lib()
    k = np.ascontiguousarray(keys, dtype=np.int64)
    c = np.ascontiguousarray(counts, dtype=np.uint32)
    ok, oc, orank = np.zeros(k.size, np.int64), np.zeros(k.size, np.uint32), np.zeros(k.size, np.uint32)
    n_out = ctypes.c_size_t(0)
    L.sor_finalize_used_list.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32,
                                         ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                         ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t)]
    rc = L.sor_finalize_used_list(k.ctypes.data, c.ctypes.data, k.size, record_count, merge_ed, min_count_fold,
                                  cells_fold_below_max, ok.ctypes.data, oc.ctypes.data, orank.ctypes.data,
                                  ctypes.byref(n_out))
    assert rc == 0
    m = n_out.value
    return ok[:m], oc[:m], orank[:m]


# ---- UMI pair distances (sor_umi.c) --------------------------------------------------------------------------
def umi_pair(w1, w2, umi_len=12):
    """w1, w2: umi_len + 2 4-bit codes each (umis/umi_length, config.xml:264: 12 as shipped)"""
    a = np.ascontiguousarray(w1, dtype=np.uint8)
    b = np.ascontiguousarray(w2, dtype=np.uint8)
    assert a.size >= umi_len + 2 and b.size >= umi_len + 2
    L = lib()
    L.sor_umi_pair_len.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    return int(L.sor_umi_pair_len(a.ctypes.data, b.ctypes.data, umi_len))


def umi_matrix(windows, umi_len=12):
    """windows: uint8 [n, umi_len + 2] 4-bit codes -> uint8 [n, n] packed (ed | pos1 << 4 | pos2 << 6)"""
    w = np.ascontiguousarray(windows, dtype=np.uint8)
    n = w.shape[0]
    assert n == 0 or w.shape[1] == umi_len + 2
    out = np.zeros((n, n), dtype=np.uint8)
    L = lib()
    L.sor_umi_matrix_len.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    L.sor_umi_matrix_len.restype = None
    L.sor_umi_matrix_len(w.ctypes.data, n, umi_len, out.ctypes.data)
    return out


def umi_window_3p(x, adapter_end, bc_end, umi_len=12):
    out = np.zeros(umi_len + 2, dtype=np.uint8)
    L = lib()
    L.sor_umi_window_3p_len.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    rc = L.sor_umi_window_3p_len(x.encode(), len(x), adapter_end, bc_end, umi_len, out.ctypes.data)
    return None if rc else out


def umi_window_5p(x, adapter_end, bc_end, umi_len=12):
    out = np.zeros(umi_len + 2, dtype=np.uint8)
    L = lib()
    L.sor_umi_window_5p_len.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    rc = L.sor_umi_window_5p_len(x.encode(), len(x), adapter_end, bc_end, umi_len, out.ctypes.data)
    return None if rc else out


def limited_compare(a, b, threshold=4):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    b = np.ascontiguousarray(b, dtype=np.uint8)
    L = lib()
    L.sor_limited_compare.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    return int(L.sor_limited_compare(a.ctypes.data, a.size, b.ctypes.data, b.size, threshold))


# ---- read-name writer (sor_name.c) ---------------------------------------------------------------------------
def format_read_name(read_name, raw_seq, raw_qual, scan, bc=None, rank=0, read_id=0, five_prime=False):
    """scan: SCAN_RESULT_DTYPE record, bc: ASSIGN_DTYPE record or None -> str, or None where the reference throws"""
    L = lib()
    L.sor_format_read_name.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p,
                                       ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32, ctypes.c_int, ctypes.c_char_p,
                                       ctypes.c_size_t]
    sc = np.zeros(1, dtype=SCAN_RESULT_DTYPE)
    sc[0] = scan
    out = ctypes.create_string_buffer(1200)
    bp = None
    if bc is not None:
        b = np.zeros(1, dtype=ASSIGN_DTYPE)
        b[0] = bc
        bp = b.ctypes.data
    n = L.sor_format_read_name(read_name.encode(), raw_seq.encode(), raw_qual.encode(), len(raw_seq), sc.ctypes.data, bp,
                               int(rank), int(read_id), int(five_prime), out, 1200)
    return None if n < 0 else out.value.decode()


def fastq_record(read_name, qual_header, raw_seq, raw_qual, scan, bc=None, rank=0, read_id=0, five_prime=False,
                 trim_fastq=False, force_failed=False):
    """the record as the pass-2 writer emits it -> (bytes, passed) or (None, passed) where the reference throws"""
    L = lib()
    L.sor_fastq_record.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int,
                                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32, ctypes.c_int, ctypes.c_int,
                                   ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_int)]
    sc = np.zeros(1, dtype=SCAN_RESULT_DTYPE)
    sc[0] = scan
    bp = None
    if bc is not None:
        b = np.zeros(1, dtype=ASSIGN_DTYPE)
        b[0] = bc
        bp = b.ctypes.data
    cap = 2 * len(raw_seq) + len(read_name) + len(qual_header) + 1400
    out = ctypes.create_string_buffer(cap)
    passed = ctypes.c_int(0)
    seq = raw_seq if isinstance(raw_seq, bytes) else raw_seq.encode()
    qual = raw_qual if isinstance(raw_qual, bytes) else raw_qual.encode()
    n = L.sor_fastq_record(read_name.encode(), qual_header.encode(), seq, qual, len(seq), sc.ctypes.data, bp, int(rank),
                           int(read_id), int(five_prime), int(trim_fastq), int(force_failed), out, cap, ctypes.byref(passed))
    return (None if n < 0 else out.raw[:n]), bool(passed.value)


def fmt_dec1(f):
    L = lib()
    L.sor_fmt_dec1.argtypes = [ctypes.c_float, ctypes.c_char_p]
    out = ctypes.create_string_buffer(64)
    L.sor_fmt_dec1(ctypes.c_float(f), out)
    return out.value.decode()


# ---- chimera splitter (sor_chimera.c) ---------------------------------------------------------------------------
class _ChimeraParams(ctypes.Structure):
    _fields_ = [("tso_complete", ctypes.c_char_p), ("adapter_complete", ctypes.c_char_p), ("tso_max_errors", ctypes.c_int32),
                ("adapter_max_errors", ctypes.c_int32), ("internal_pat_len", ctypes.c_int32),
                ("internal_pat_frac", ctypes.c_float), ("window_polya", ctypes.c_int32), ("bc_umi_len", ctypes.c_int32)]


class _ChimeraResult(ctypes.Structure):
    _fields_ = [("n_split", ctypes.c_int32), ("pos", ctypes.c_int32 * 2), ("reason", ctypes.c_int32 * 2),
                ("multi_chimeric", ctypes.c_int32), ("n_matches", ctypes.c_int32)]


SPLIT_REASONS = ["REV_ADAPTER", "FWD_ADAPTER", "REV_ADAPTER_FWD_ADAPTER", "REV_ADAPTER_FWD_TSO", "REV_TSO_FWD_ADAPTER",
                 "REV_TSO_FWD_TSO", "READSTART"]


def chimera_params(tso="AAGCAGTGGTATCAACGCAGAGTACAT", adapter="CTACACGACGCTCTTCCGATCT", tso_max=6, adapter_max=5, bc_umi=28):
    return _ChimeraParams(tso.encode(), adapter.encode(), tso_max, adapter_max, 15, 0.70, 150, bc_umi)


def chimera_split(read, params=None):
    """-> (rc, [(reason name, pos)], multi_chimeric, n_matches, raw result)"""
    L = lib()
    p = chimera_params() if params is None else params
    r = _ChimeraResult()
    L.sor_chimera_split.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    rc = L.sor_chimera_split(read.encode(), len(read), ctypes.byref(p), ctypes.byref(r))
    return rc, [(SPLIT_REASONS[r.reason[i]], r.pos[i]) for i in range(r.n_split)], bool(r.multi_chimeric), r.n_matches, r


def chimera_fragment_name(name, raw_result, fragment):
    L = lib()
    L.sor_chimera_fragment_name.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t]
    out = ctypes.create_string_buffer(len(name) + 64)
    n = L.sor_chimera_fragment_name(name.encode(), ctypes.byref(raw_result), fragment, out, len(name) + 64)
    return out.value.decode() if n >= 0 else None


# ---- UMI clustering (sor_cluster.c) --------------------------------------------------------------------------------
UMI_ASSIGNMENT_DTYPE = np.dtype([("center", "<i4"), ("offset", "i1"), ("ed", "i1"), ("ed_second", "i1"), ("pos2", "i1")])
UMI_CLUSTER_PARAMS_DTYPE = np.dtype([("complete_link_ed", "<i4"), ("single_link_ed", "<i4"), ("single_link_switch", "<i4"),
                                     ("fold_depth_below_max", "<i4"), ("own_clusterer_above", "<i4")])


def umi_cluster_params(complete_ed=2, single_ed=1, single_switch=3000, fold=50, own_above=100):
    p = np.zeros(1, dtype=UMI_CLUSTER_PARAMS_DTYPE)
    p[0] = (complete_ed, single_ed, single_switch, fold, own_above)
    return p


def umi_cluster_group(mat, n, mean_qv, params=None):
    """mat: uint8 [n*n] packed; -> (assignments structured array [n], skipped bool [n])"""
    L = lib()
    p = umi_cluster_params() if params is None else params
    m = np.ascontiguousarray(mat, dtype=np.uint8)
    q = np.ascontiguousarray(mean_qv, dtype=np.float32)
    out = np.zeros(max(n, 1), dtype=UMI_ASSIGNMENT_DTYPE)
    sk = np.zeros(max(n, 1), dtype=np.uint8)
    L.sor_umi_cluster_group.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.c_void_p]
    L.sor_umi_cluster_group(m.ctypes.data, n, q.ctypes.data, p.ctypes.data, out.ctypes.data, sk.ctypes.data)
    return out[:n], sk[:n].astype(bool)


# ---- genomic-region grouping (sor_group.c) ------------------------------------------------------------------------
def region_group(pos, reverse, max_dist=500, keep_data_end=False):
    """pos: list with None for absent; -> (region list, n_done)"""
    L = lib()
    n = len(pos)
    p = np.array([0 if v is None else v for v in pos], dtype=np.int32)
    h = np.array([v is not None for v in pos], dtype=np.uint8)
    r = np.ascontiguousarray(reverse, dtype=np.uint8)
    out = np.zeros(max(n, 1), dtype=np.int32)
    nd = ctypes.c_int32(0)
    L.sor_region_group.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int32, ctypes.c_int32, ctypes.c_int, ctypes.c_void_p,
                                                           ctypes.c_void_p]
    L.sor_region_group(p.ctypes.data, h.ctypes.data, r.ctypes.data, n, max_dist, int(keep_data_end), out.ctypes.data,
                       ctypes.byref(nd))
    return out[:n].tolist(), nd.value


CIGAR_OPS = "MIDNSHP=X"


def ref_position_at_read_position(cigar, alignment_start, position):
    L = lib()
    c = np.array([(ln << 4) | CIGAR_OPS.index(op) for op, ln in cigar], dtype=np.uint32)
    out = ctypes.c_int32(0)
    L.sor_ref_position_at_read_position.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int32, ctypes.c_int32,
                                                    ctypes.c_void_p]
    rc = L.sor_ref_position_at_read_position(c.ctypes.data, c.size, alignment_start, position, ctypes.byref(out))
    return out.value if rc == 1 else None
