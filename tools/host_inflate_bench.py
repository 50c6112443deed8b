"""Rate of the library's host gzip decoder (smi_gz_inflate_into) on FASTQ text, beside zlib (Python's zlib module, what run_files used
before) and, when the image has it, libdeflate.so.0 -- one thread and N threads (one file per thread, as run_files does).
-> one JSON line (profiles/r03/host_inflate.json).  CPU only."""
import ctypes
import json
import os
import sys
import time
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft  # noqa: E402


def fastq(rng, n):
    recs = []
    for i in range(n):
        ln = int(rng.integers(448, 1948))
        s = bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, ln)])
        q = bytes((rng.integers(35, 64, ln) + 33).astype(np.uint8))
        recs.append(b"@read%d_some_name\n" % i + s + b"\n+\n" + q + b"\n")
    return b"".join(recs)


def main():
    import importlib

    graft.load_package()
    lib = importlib.import_module(graft.PKG_NAME + ".lib")
    n_threads = int(os.environ.get("SMI_HIB_THREADS", "16"))
    rng = np.random.default_rng(1)
    text = fastq(rng, 20000)
    ld = None
    try:
        ld = ctypes.CDLL("libdeflate.so.0")
        ld.libdeflate_alloc_decompressor.restype = ctypes.c_void_p
        ld.libdeflate_gzip_decompress_ex.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t,
                                                     ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]
    except OSError:
        pass
    res = {"text_bytes": len(text), "threads": n_threads, "streams": {}}
    for name, level in (("zlib level 1", 1), ("zlib level 6", 6)):
        c = zlib.compressobj(level, zlib.DEFLATED, 31)
        gz = np.frombuffer(c.compress(text) + c.flush(), dtype=np.uint8)

        def own(_k=0):
            return lib.gz_inflate(gz)

        def py_zlib(_k=0):
            return zlib.decompress(gz.tobytes(), 31, len(text))

        def libdeflate(_k=0):
            d = ld.libdeflate_alloc_decompressor()
            out = np.empty(len(text), dtype=np.uint8)
            ai, ao = ctypes.c_size_t(0), ctypes.c_size_t(0)
            ld.libdeflate_gzip_decompress_ex(d, gz.ctypes.data, gz.size, out.ctypes.data, out.size, ctypes.byref(ai), ctypes.byref(ao))
            return out

        assert own().tobytes() == text
        row = {"ratio": round(len(text) / gz.size, 3)}
        for label, fn in (("own", own), ("python_zlib", py_zlib)) + ((("libdeflate", libdeflate),) if ld else ()):
            best = 1e9
            for _ in range(5):
                t = time.perf_counter()
                fn()
                best = min(best, time.perf_counter() - t)
            row[label + "_MBps_1_thread"] = round(len(text) / best / 1e6, 1)
            with ThreadPoolExecutor(n_threads) as pool:
                list(pool.map(fn, range(n_threads)))
                t = time.perf_counter()
                list(pool.map(fn, range(4 * n_threads)))
                dt = time.perf_counter() - t
            row[label + f"_MBps_{n_threads}_threads"] = round(4 * n_threads * len(text) / dt / 1e6, 1)
        res["streams"][name] = row
    print(json.dumps(res))


if __name__ == "__main__":
    main()
