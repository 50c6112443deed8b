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
