"""tools/jvm_exec.py (the bytecode interpreter behind tests/golden/ref_exec_*.json) checked against answers that do NOT come from this
repository: (1) third-party pure-Java methods from jars the reference ships, executed by the interpreter, against PUBLISHED test vectors
(CRC-32C: RFC 3720 B.4; MurmurHash3 x86_32: the SMHasher / reference-implementation vectors; xxHash32: the xxHash specification;
Levenshtein: the textbook pairs; greatest common divisors); (2) single instructions assembled into a class file by this test, against the
Java Virtual Machine Specification's own statements about them (shift-count masking, float-to-integer saturation, NaN compares, the signs
of idiv / irem, narrowing conversions, long compare, integer overflow wrap).  It does not make the runtime a JVM; it removes the doubt that
interpreter and oracle share one author's misreading of these rules."""
import io
import math
import os
import struct
import sys
import zipfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
LIBDIR = "/root/reference/Jar/lib"
pytestmark = pytest.mark.skipif(not os.path.isdir(LIBDIR), reason="the reference jars exist only in the build container")


def _jvm(extra_zip=None):
    import jvm_exec

    jars = [os.path.join(LIBDIR, j) for j in ("commons-codec-1.17.0.jar", "commons-lang3-3.17.0.jar", "commons-math3-3.6.1.jar")]
    j = jvm_exec.JVM(jars)
    # the JDK intrinsics the library methods below call (plain bit operations, as the JDK API documents them)
    rot = lambda v, n, bits: ((v << (n % bits)) | ((v & (2 ** bits - 1)) >> (bits - n % bits))) & (2 ** bits - 1)  # noqa: E731
    j.natives.setdefault("java/lang/Integer.rotateLeft:(II)I", lambda jv, v, n: jvm_exec.i32(rot(v & 0xFFFFFFFF, n & 31, 32) if n & 31 else v))
    j.natives.setdefault("java/lang/Long.rotateLeft:(JI)J", lambda jv, v, n: jvm_exec.i64(rot(v & (2 ** 64 - 1), n & 63, 64) if n & 63 else v))
    ntz = lambda v, bits: bits if v == 0 else (v & -v).bit_length() - 1  # noqa: E731
    j.natives.setdefault("java/lang/Integer.numberOfTrailingZeros:(I)I", lambda jv, v: ntz(v & 0xFFFFFFFF, 32))
    j.natives.setdefault("java/lang/Long.numberOfTrailingZeros:(J)I", lambda jv, v: ntz(v & (2 ** 64 - 1), 64))
    if extra_zip is not None:
        z = zipfile.ZipFile(io.BytesIO(extra_zip))
        j.zips.append(z)
        for n in z.namelist():
            j.index.setdefault(n[:-6], z)
    return j, jvm_exec


def _bytes(jx, data):
    return jx.JArray("B", [b - 256 if b > 127 else b for b in data])


def test_crc32c_rfc3720_vectors():
    j, jx = _jvm()
    cases = [(bytes(32), 0x8A9136AA), (b"\xff" * 32, 0x62A8AB43), (bytes(range(32)), 0x46DD794E), (bytes(range(31, -1, -1)), 0x113FDB5C),
             (b"123456789", 0xE3069283)]
    for data, want in cases:
        o = j.new("org/apache/commons/codec/digest/PureJavaCrc32C")
        j.call_virtual(o, "update", "([BII)V", _bytes(jx, data), 0, len(data))
        assert j.call_virtual(o, "getValue", "()J") & 0xFFFFFFFF == want


def test_murmur3_x86_32_published_vectors():
    j, jx = _jvm()
    cases = [(b"", 0, 0x00000000), (b"", 1, 0x514E28B7), (b"", 0xFFFFFFFF, 0x81F16F39), (b"\xff\xff\xff\xff", 0, 0x76293B50),
             (b"\x21\x43\x65\x87", 0, 0xF55B516B), (b"\x21\x43\x65\x87", 0x5082EDEE, 0x2362F9DE), (b"\x21\x43\x65", 0, 0x7E4A8634),
             (b"\x21\x43", 0, 0xA0F7B07A), (b"\x21", 0, 0x72661CF4), (b"\0\0\0\0", 0, 0x2362F9DE), (b"abc", 0, 0xB3DD93FA),
             (b"Hello, world!", 1234, 0xFAF6CDB3), (b"The quick brown fox jumps over the lazy dog", 0x9747B28C, 0x2FA826CD)]
    for data, seed, want in cases:
        got = j.call_static("org/apache/commons/codec/digest/MurmurHash3", "hash32x86", "([BIII)I", _bytes(jx, data), 0, len(data), jx.i32(seed))
        assert got & 0xFFFFFFFF == want, (data, seed)


def test_xxhash32_specification_vectors():
    j, jx = _jvm()
    for data, want in ((b"", 0x02CC5D05), (b"a", 0x550D7456), (b"abc", 0x32D153FF), (b"Nobody inspects the spammish repetition", 0xE2293B2F)):
        o = j.new("org/apache/commons/codec/digest/XXHash32")
        j.call_virtual(o, "update", "([BII)V", _bytes(jx, data), 0, len(data))
        assert j.call_virtual(o, "getValue", "()J") & 0xFFFFFFFF == want, data


def test_levenshtein_and_gcd():
    j, jx = _jvm()
    for a, b, want in (("kitten", "sitting", 3), ("flaw", "lawn", 2), ("", "abc", 3), ("intention", "execution", 5), ("same", "same", 0)):
        o = j.call_static("org/apache/commons/lang3/StringUtils", "getLevenshteinDistance", "(Ljava/lang/CharSequence;Ljava/lang/CharSequence;)I", a, b)
        assert o == want
    for num, den, want in ((1071, 462, (51, 22)), (-6, 8, (-3, 4)), (2 ** 30, 2 ** 20 * 3, (1024, 3)), (0, 7, (0, 1))):
        fr = j.call_static("org/apache/commons/lang3/math/Fraction", "getReducedFraction", "(II)Lorg/apache/commons/lang3/math/Fraction;", num, den)
        assert (fr.f["numerator"], fr.f["denominator"]) == want


# ---- single instructions, assembled here ---------------------------------------------------------------------------------------------
def _class_file(name, methods):
    """a minimal class file: public static methods [(name, descriptor, max_stack, max_locals, code bytes)]"""
    cp = [None]

    def utf8(s):
        cp.append(b"\x01" + struct.pack(">H", len(s)) + s.encode())
        return len(cp) - 1

    def cls(s):
        i = utf8(s)
        cp.append(b"\x07" + struct.pack(">H", i))
        return len(cp) - 1

    this, sup, code_attr = cls(name), cls("java/lang/Object"), utf8("Code")
    body = b""
    for mname, desc, ms, ml, code in methods:
        attr = struct.pack(">HHI", ms, ml, len(code)) + code + struct.pack(">HH", 0, 0)
        body += struct.pack(">HHHH", 0x0009, utf8(mname), utf8(desc), 1) + struct.pack(">HI", code_attr, len(attr)) + attr
    pool = b"".join(cp[1:])
    return (b"\xca\xfe\xba\xbe" + struct.pack(">HH", 0, 52) + struct.pack(">H", len(cp)) + pool + struct.pack(">HHH", 0x0021, this, sup) +
            struct.pack(">HH", 0, 0) + struct.pack(">H", len(methods)) + body + struct.pack(">H", 0))


ILOAD0, ILOAD1, LLOAD0, LLOAD2, FLOAD0, FLOAD1, DLOAD0, DLOAD2 = b"\x1a", b"\x1b", b"\x1e", b"\x20", b"\x22", b"\x23", b"\x26", b"\x28"
IRET, LRET, FRET, DRET = b"\xac", b"\xad", b"\xae", b"\xaf"
OPS = [  # name, descriptor, stack, locals, code
    ("ishl", "(II)I", 2, 2, ILOAD0 + ILOAD1 + b"\x78" + IRET), ("ishr", "(II)I", 2, 2, ILOAD0 + ILOAD1 + b"\x7a" + IRET),
    ("iushr", "(II)I", 2, 2, ILOAD0 + ILOAD1 + b"\x7c" + IRET), ("lshl", "(JI)J", 3, 3, LLOAD0 + b"\x1c" + b"\x79" + LRET),
    ("lshr", "(JI)J", 3, 3, LLOAD0 + b"\x1c" + b"\x7b" + LRET), ("lushr", "(JI)J", 3, 3, LLOAD0 + b"\x1c" + b"\x7d" + LRET),
    ("idiv", "(II)I", 2, 2, ILOAD0 + ILOAD1 + b"\x6c" + IRET), ("irem", "(II)I", 2, 2, ILOAD0 + ILOAD1 + b"\x70" + IRET),
    ("ldiv", "(JJ)J", 4, 4, LLOAD0 + LLOAD2 + b"\x6d" + LRET), ("lrem", "(JJ)J", 4, 4, LLOAD0 + LLOAD2 + b"\x71" + LRET),
    ("iadd", "(II)I", 2, 2, ILOAD0 + ILOAD1 + b"\x60" + IRET), ("imul", "(II)I", 2, 2, ILOAD0 + ILOAD1 + b"\x68" + IRET),
    ("lmul", "(JJ)J", 4, 4, LLOAD0 + LLOAD2 + b"\x69" + LRET), ("ineg", "(I)I", 1, 1, ILOAD0 + b"\x74" + IRET),
    ("f2i", "(F)I", 1, 1, FLOAD0 + b"\x8b" + IRET), ("f2l", "(F)J", 2, 1, FLOAD0 + b"\x8c" + LRET), ("d2i", "(D)I", 2, 2, DLOAD0 + b"\x8e" + IRET),
    ("d2l", "(D)J", 2, 2, DLOAD0 + b"\x8f" + LRET), ("i2b", "(I)I", 1, 1, ILOAD0 + b"\x91" + IRET), ("i2c", "(I)I", 1, 1, ILOAD0 + b"\x92" + IRET),
    ("i2s", "(I)I", 1, 1, ILOAD0 + b"\x93" + IRET), ("l2i", "(J)I", 2, 2, LLOAD0 + b"\x88" + IRET), ("i2f", "(I)F", 1, 1, ILOAD0 + b"\x86" + FRET),
    ("l2f", "(J)F", 2, 2, LLOAD0 + b"\x89" + FRET), ("d2f", "(D)F", 2, 2, DLOAD0 + b"\x90" + FRET),
    ("fcmpl", "(FF)I", 2, 2, FLOAD0 + FLOAD1 + b"\x95" + IRET), ("fcmpg", "(FF)I", 2, 2, FLOAD0 + FLOAD1 + b"\x96" + IRET),
    ("dcmpl", "(DD)I", 4, 4, DLOAD0 + DLOAD2 + b"\x97" + IRET), ("dcmpg", "(DD)I", 4, 4, DLOAD0 + DLOAD2 + b"\x98" + IRET),
    ("lcmp", "(JJ)I", 4, 4, LLOAD0 + LLOAD2 + b"\x94" + IRET), ("fmul", "(FF)F", 2, 2, FLOAD0 + FLOAD1 + b"\x6a" + FRET),
    ("fadd", "(FF)F", 2, 2, FLOAD0 + FLOAD1 + b"\x62" + FRET), ("frem", "(FF)F", 2, 2, FLOAD0 + FLOAD1 + b"\x72" + FRET),
]


@pytest.fixture(scope="module")
def ops():
    buf = io.BytesIO()
    with zipfile.ZipFile(buf, "w") as z:
        z.writestr("t/Ops.class", _class_file("t/Ops", OPS))
    j, jx = _jvm(buf.getvalue())
    desc = {n: d for n, d, *_ in OPS}
    return lambda name, *a: j.call_static("t/Ops", name, desc[name], *a)


IMIN, IMAX, LMIN, LMAX = -2 ** 31, 2 ** 31 - 1, -2 ** 63, 2 ** 63 - 1


def test_shift_counts_are_masked(ops):
    """JVMS 6.5 ishl / lshl ...: only the low five (int) or six (long) bits of the count are used"""
    assert ops("ishl", 1, 33) == 2 and ops("ishl", 1, 32) == 1 and ops("ishl", 1, -1) == IMIN and ops("ishl", 3, 31) == IMIN
    assert ops("ishr", -8, 33) == -4 and ops("ishr", IMIN, 31) == -1 and ops("iushr", -1, 28) == 15 and ops("iushr", -8, 32) == -8
    assert ops("lshl", 1, 65) == 2 and ops("lshl", 1, 64) == 1 and ops("lshl", 1, 63) == LMIN and ops("lshl", 1, -1) == LMIN
    assert ops("lshr", -16, 66) == -4 and ops("lushr", -1, 60) == 15 and ops("lushr", -1, 64) == -1 and ops("lushr", LMIN, 63) == 1


def test_integer_division_signs_and_wrap(ops):
    """idiv truncates towards zero, irem takes the dividend's sign, MIN / -1 wraps (JVMS 6.5 idiv, irem)"""
    assert ops("idiv", -7, 2) == -3 and ops("idiv", 7, -2) == -3 and ops("irem", -7, 2) == -1 and ops("irem", 7, -2) == 1
    assert ops("idiv", IMIN, -1) == IMIN and ops("irem", IMIN, -1) == 0
    assert ops("ldiv", -7, 2) == -3 and ops("lrem", -7, 2) == -1 and ops("ldiv", LMIN, -1) == LMIN and ops("lrem", LMIN, -1) == 0
    assert ops("iadd", IMAX, 1) == IMIN and ops("imul", 65536, 65536) == 0 and ops("imul", 46341, 46341) == -2147479015
    assert ops("lmul", 2 ** 32, 2 ** 32) == 0 and ops("lmul", LMAX, 2) == -2 and ops("ineg", IMIN) == IMIN


def test_float_to_integer_conversions_saturate(ops):
    """NaN -> 0, out-of-range values -> MIN / MAX, otherwise round towards zero (JVMS 6.5 f2i, f2l, d2i, d2l)"""
    nan, inf = float("nan"), float("inf")
    assert ops("f2i", nan) == 0 and ops("f2i", inf) == IMAX and ops("f2i", -inf) == IMIN and ops("f2i", 3e9) == IMAX and ops("f2i", -2.9) == -2
    assert ops("f2l", nan) == 0 and ops("f2l", 1e30) == LMAX and ops("f2l", -1e30) == LMIN and ops("f2l", 2.5) == 2
    assert ops("d2i", nan) == 0 and ops("d2i", 1e10) == IMAX and ops("d2i", -1e10) == IMIN and ops("d2i", -0.9) == 0 and ops("d2i", 2147483647.9) == IMAX
    assert ops("d2l", nan) == 0 and ops("d2l", 1e19) == LMAX and ops("d2l", -1e19) == LMIN and ops("d2l", 9007199254740993.0) == 9007199254740992


def test_narrowing_and_widening(ops):
    assert ops("i2b", 128) == -128 and ops("i2b", 255) == -1 and ops("i2b", -129) == 127
    assert ops("i2c", -1) == 65535 and ops("i2c", 65536) == 0 and ops("i2s", 32768) == -32768 and ops("i2s", 65535) == -1
    assert ops("l2i", 2 ** 32 + 5) == 5 and ops("l2i", 2 ** 31) == IMIN
    assert ops("i2f", 16777217) == 16777216.0 and ops("i2f", IMAX) == 2147483648.0 and ops("l2f", LMAX) == 9.223372036854775807e18
    assert ops("d2f", 1e40) == float("inf") and ops("d2f", 0.1) == struct.unpack("f", struct.pack("f", 0.1))[0]


def test_compares_with_nan(ops):
    """fcmpl pushes -1 and fcmpg +1 when either operand is NaN (JVMS 6.5 fcmp<op>); lcmp is a three-way compare"""
    nan = float("nan")
    assert ops("fcmpl", nan, 1.0) == -1 and ops("fcmpg", nan, 1.0) == 1 and ops("fcmpl", 1.0, nan) == -1 and ops("fcmpg", 1.0, nan) == 1
    assert ops("dcmpl", nan, nan) == -1 and ops("dcmpg", nan, nan) == 1
    assert ops("fcmpl", 1.0, 2.0) == -1 and ops("fcmpg", 2.0, 1.0) == 1 and ops("fcmpl", 0.0, -0.0) == 0 and ops("dcmpg", 3.0, 3.0) == 0
    assert ops("lcmp", LMIN, LMAX) == -1 and ops("lcmp", 5, 5) == 0 and ops("lcmp", LMAX, LMIN) == 1


def test_float_arithmetic_is_single_precision(ops):
    """fmul / fadd round to float (not double): 16777216f + 1f == 16777216f; 0.1f * 3f is the float product"""
    f = lambda x: struct.unpack("f", struct.pack("f", x))[0]  # noqa: E731
    assert ops("fadd", 16777216.0, 1.0) == 16777216.0
    assert ops("fmul", f(0.1), 3.0) == f(f(0.1) * 3.0)
    assert ops("frem", 5.5, 2.0) == 1.5 and ops("frem", -5.5, 2.0) == -1.5      # the sign of the dividend (fmod, not IEEE remainder)
