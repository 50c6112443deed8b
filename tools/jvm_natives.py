"""JDK natives of tools/jvm_exec.py (test infrastructure; see the tiers in that file's header).

Only behaviour the Java SE specification fixes exactly is supplied.  Nothing with a hash-ordered iteration is here."""
import math

import numpy as np

from jvm_exec import JArray, JBox, JLambda, JObject, JavaThrow, Unsupported, f32, i32, i64

TIER_A = {"java/lang/Object.<init>:()V"}


def tier_of(key):
    if key in TIER_A:
        return "A"
    if key.startswith("$hash-iteration(jdk"):
        return "D"
    if key.startswith(("java/util/", "it/unimi/", "org/eclipse/")):
        return "C"
    return "B"


def jhash_str(s):
    h = 0
    for ch in s:
        h = (31 * h + ord(ch)) & 0xFFFFFFFF
    return i32(h)


def float_to_string(v, is_float):
    """Float.toString / Double.toString: shortest decimal that round-trips, Java's layout (JLS: computerised scientific
    notation outside [1e-3, 1e7))"""
    if v != v:
        return "NaN"
    if v in (math.inf, -math.inf):
        return "Infinity" if v > 0 else "-Infinity"
    if v == 0:
        return "-0.0" if math.copysign(1.0, v) < 0 else "0.0"
    a = abs(v)
    digits = np.format_float_scientific(np.float32(a) if is_float else np.float64(a), unique=True, trim="-")
    mant, exp = digits.split("e")
    exp = int(exp)
    ds = mant.replace(".", "")
    sign = "-" if v < 0 else ""
    if 1e-3 <= a < 1e7:
        if exp >= 0:
            ip = ds[:exp + 1].ljust(exp + 1, "0")
            fp = ds[exp + 1:] or "0"
        else:
            ip = "0"
            fp = "0" * (-exp - 1) + ds
        return f"{sign}{ip}.{fp}"
    return f"{sign}{ds[0]}.{ds[1:] or '0'}E{exp}"


def java_round(v, is_float):
    """Math.round: floor(x + 1/2) with the exact-tie rule of the JLS (Java 7+)"""
    if v != v:
        return 0
    r = math.floor(v + 0.5) if abs(v) < 2 ** 52 else v
    if is_float:
        return max(-2 ** 31, min(2 ** 31 - 1, int(r)))
    return max(-2 ** 63, min(2 ** 63 - 1, int(r)))


class Uninit:
    """reference pushed by `new java/lang/String` (and boxes) until <init> supplies the immutable value"""
    __slots__ = ("cls",)

    def __init__(self, cls):
        self.cls = cls


def _npe(jvm):
    jvm.throw("java/lang/NullPointerException")


def _sb(o):
    return o.native


def install(jvm):
    N = jvm.natives
    N["$float_to_string"] = float_to_string

    # ---- logging has no effect on results: loggers are inert objects
    def logger(j, *a):
        o = JObject("$Logger")
        return o

    N["org/apache/logging/log4j/LogManager.getLogger"] = logger
    N["$Logger.*"] = lambda j, o, *a: None
    N["org/apache/logging/log4j/Logger.*"] = lambda j, o, *a: None

    # ---- java.lang.Object ------------------------------------------------------------------------------------------
    N["java/lang/Object.<init>:()V"] = lambda j, o: None
    N["java/lang/Object.getClass"] = lambda j, o: j.class_object(j.class_of(o))
    N["java/lang/Object.equals"] = lambda j, a, b: 1 if (a is b or (isinstance(a, (str, JBox)) and a == b)) else 0
    N["java/lang/Class.getSimpleName"] = lambda j, c: c.native.split("/")[-1].split("$")[-1]
    N["java/lang/Class.getName"] = lambda j, c: c.native.replace("/", ".")
    N["java/lang/Class.desiredAssertionStatus"] = lambda j, c: 0

    # ---- java.lang.String ------------------------------------------------------------------------------------------
    def s_init(j, o, *a):
        if not a:
            return ""
        v = a[0]
        if isinstance(v, str):
            return v
        if isinstance(v, JArray):
            if len(a) == 3 and isinstance(a[1], int) and isinstance(a[2], int):
                seg = v.a[a[1]:a[1] + a[2]]
            else:
                seg = v.a
            if v.etype == "C":
                return "".join(chr(c) for c in seg)
            return bytes(b & 255 for b in seg).decode("latin-1")  # ASCII payloads only (bases, qualities)
        if isinstance(v, JObject) and v.cls == "java/lang/StringBuilder":
            return "".join(v.native)
        raise Unsupported("String.<init> variant")

    N["java/lang/String.<init>"] = s_init

    def s_char_at(j, s, i):
        if not 0 <= i < len(s):
            j.throw("java/lang/StringIndexOutOfBoundsException", f"index {i}, length {len(s)}")
        return ord(s[i])

    def s_substring(j, s, b, e=None):
        e = len(s) if e is None else e
        if b < 0 or e > len(s) or b > e:
            j.throw("java/lang/StringIndexOutOfBoundsException", f"begin {b}, end {e}, length {len(s)}")
        return s[b:e]

    N["java/lang/String.length"] = lambda j, s: len(s)
    N["java/lang/String.isEmpty"] = lambda j, s: 1 if not s else 0
    N["java/lang/String.charAt"] = s_char_at
    N["java/lang/String.substring"] = s_substring
    N["java/lang/String.subSequence"] = s_substring
    N["java/lang/String.equals"] = lambda j, s, o: 1 if isinstance(o, str) and o == s else 0
    N["java/lang/String.equalsIgnoreCase"] = lambda j, s, o: 1 if isinstance(o, str) and o.lower() == s.lower() else 0
    N["java/lang/String.hashCode"] = lambda j, s: jhash_str(s)
    N["java/lang/String.toString"] = lambda j, s: s
    N["java/lang/String.toCharArray"] = lambda j, s: JArray("C", [ord(c) for c in s])
    N["java/lang/String.getBytes"] = lambda j, s, *a: JArray("B", [((b + 128) & 255) - 128 for b in s.encode("latin-1")])
    def s_get_bytes_into(j, s, b, e, dst, d0):
        if b < 0 or e > len(s) or b > e or d0 < 0 or d0 + (e - b) > len(dst.a):
            j.throw("java/lang/StringIndexOutOfBoundsException")
        for k in range(b, e):
            dst.a[d0 + k - b] = ((ord(s[k]) + 128) & 255) - 128

    N["java/lang/String.getBytes:(II[BI)V"] = s_get_bytes_into
    N["java/lang/String.getChars:(II[CI)V"] = lambda j, s, b, e, dst, d0: dst.a.__setitem__(slice(d0, d0 + e - b), [ord(c) for c in s[b:e]])
    N["java/lang/String.indexOf:(I)I"] = lambda j, s, c: s.find(chr(c))
    N["java/lang/String.indexOf:(II)I"] = lambda j, s, c, f: s.find(chr(c), max(f, 0))
    N["java/lang/String.indexOf:(Ljava/lang/String;)I"] = lambda j, s, t: s.find(t)
    N["java/lang/String.indexOf:(Ljava/lang/String;I)I"] = lambda j, s, t, f: s.find(t, max(f, 0))
    N["java/lang/String.lastIndexOf:(I)I"] = lambda j, s, c: s.rfind(chr(c))
    N["java/lang/String.lastIndexOf:(Ljava/lang/String;)I"] = lambda j, s, t: s.rfind(t)
    N["java/lang/String.contains"] = lambda j, s, t: 1 if j.to_jstring(t) in s else 0
    N["java/lang/String.startsWith:(Ljava/lang/String;)Z"] = lambda j, s, t: 1 if s.startswith(t) else 0
    N["java/lang/String.endsWith"] = lambda j, s, t: 1 if s.endswith(t) else 0
    N["java/lang/String.concat"] = lambda j, s, t: s + t
    N["java/lang/String.intern"] = lambda j, s: s
    # an opaque object: static initialisers may compile patterns the executed path never uses; any use of one is Unsupported
    N["java/util/regex/Pattern.compile:(Ljava/lang/String;)Ljava/util/regex/Pattern;"] = lambda j, s: JObject("java/util/regex/Pattern")
    N["java/lang/Boolean.parseBoolean:(Ljava/lang/String;)Z"] = lambda j, s: 1 if s is not None and s.lower() == "true" else 0
    N["java/lang/String.trim"] = lambda j, s: s.strip(" \t\n\r\x0b\x0c" + "".join(chr(c) for c in range(0, 33)))
    N["java/lang/String.toUpperCase:()Ljava/lang/String;"] = lambda j, s: s.upper()
    N["java/lang/String.toLowerCase:()Ljava/lang/String;"] = lambda j, s: s.lower()
    N["java/lang/String.replace:(CC)Ljava/lang/String;"] = lambda j, s, a, b: s.replace(chr(a), chr(b))
    N["java/lang/String.replace:(Ljava/lang/CharSequence;Ljava/lang/CharSequence;)Ljava/lang/String;"] = \
        lambda j, s, a, b: s.replace(j.to_jstring(a), j.to_jstring(b))
    N["java/lang/String.compareTo:(Ljava/lang/String;)I"] = lambda j, s, t: _str_cmp(s, t)
    N["java/lang/String.compareTo:(Ljava/lang/Object;)I"] = lambda j, s, t: _str_cmp(s, t)
    N["java/lang/String.valueOf:(I)Ljava/lang/String;"] = lambda j, v: str(v)
    N["java/lang/String.valueOf:(J)Ljava/lang/String;"] = lambda j, v: str(v)
    N["java/lang/String.valueOf:(C)Ljava/lang/String;"] = lambda j, v: chr(v)
    N["java/lang/String.valueOf:(Z)Ljava/lang/String;"] = lambda j, v: "true" if v else "false"
    N["java/lang/String.valueOf:(F)Ljava/lang/String;"] = lambda j, v: float_to_string(v, True)
    N["java/lang/String.valueOf:(D)Ljava/lang/String;"] = lambda j, v: float_to_string(v, False)
    N["java/lang/String.valueOf:(Ljava/lang/Object;)Ljava/lang/String;"] = lambda j, v: j.to_jstring(v)
    N["java/lang/String.valueOf:([C)Ljava/lang/String;"] = lambda j, v: "".join(chr(c) for c in v.a)
    N["java/lang/String.copyValueOf:([C)Ljava/lang/String;"] = lambda j, v: "".join(chr(c) for c in v.a)
    def s_split(j, s, rx, limit=0):
        import re

        if limit != 0:
            raise Unsupported("String.split with a limit")
        if any(ch in rx for ch in ".$|()[{^?*+\\"):
            simple = {"\\s+": r"\s+", "\\|": r"\|", "\\t": "\t", "\\.": r"\."}
            if rx not in simple:
                raise Unsupported(f"String.split regex {rx!r}")
            parts = re.split(simple[rx], s)
        else:
            parts = s.split(rx) if rx else list(s)
        if s == "":
            return JArray("Ljava/lang/String;", [""])
        while parts and parts[-1] == "":
            parts.pop()
        return JArray("Ljava/lang/String;", parts)

    N["java/lang/String.split:(Ljava/lang/String;)[Ljava/lang/String;"] = s_split

    def s_replace_first(j, s, rx, repl):
        if any(ch in rx for ch in ".$|()[{^?*+\\") or any(ch in repl for ch in "$\\"):
            raise Unsupported(f"String.replaceFirst regex {rx!r} / replacement {repl!r}")
        return s.replace(rx, repl, 1)

    N["java/lang/String.replaceFirst"] = s_replace_first

    def s_replace_all(j, s, rx, repl):
        """String.replaceAll for patterns without a regex metacharacter (a literal) and replacements without $ / backslash"""
        if any(ch in rx for ch in ".$|()[{^?*+\\") or any(ch in repl for ch in "$\\") or rx == "":
            raise Unsupported(f"String.replaceAll regex {rx!r} / replacement {repl!r}")
        return s.replace(rx, repl)

    N["java/lang/String.replaceAll"] = s_replace_all

    def s_format(j, fmt, args):
        """String.format with %s / %d / %n only (messages of exceptions and logs)"""
        import re as _re

        vals = [j.to_jstring(a) if not isinstance(a, JBox) else str(a.v) for a in args.a]
        if _re.sub(r"%[sdn%]", "", fmt).count("%"):
            raise Unsupported(f"String.format {fmt!r}")
        it = iter(vals)
        return _re.sub(r"%[sdn%]", lambda m_: "\n" if m_.group() == "%n" else "%" if m_.group() == "%%" else next(it), fmt)

    N["java/lang/String.format:(Ljava/lang/String;[Ljava/lang/Object;)Ljava/lang/String;"] = s_format
    N["java/lang/CharSequence.length"] = lambda j, s: len(j.to_jstring(s))
    N["java/lang/CharSequence.charAt"] = lambda j, s, i: s_char_at(j, j.to_jstring(s), i)
    N["java/lang/CharSequence.toString"] = lambda j, s: j.to_jstring(s)

    def _str_cmp(s, t):
        for a, b in zip(s, t):
            if a != b:
                return ord(a) - ord(b)
        return len(s) - len(t)

    # ---- java.lang.StringBuilder (a Python list of characters) --------------------------------------------------------
    def sb_new(j):
        o = JObject("java/lang/StringBuilder")
        o.native = []
        return o

    def sb_init(j, o, *a):
        if a and isinstance(a[0], str):
            o.native = list(a[0])
        return None

    def sb_append(desc):
        d = desc

        def f(j, o, v, *rest):
            if d == "[C" and rest:
                o.native.extend(chr(c) for c in v.a[rest[0]:rest[0] + rest[1]])
            elif d == "[C":
                o.native.extend(chr(c) for c in v.a)
            elif d == "L" and rest:  # append(CharSequence, start, end)
                o.native.extend(j.to_jstring(v)[rest[0]:rest[1]])
            else:
                o.native.extend(j.to_jstring(v, d if d != "L" else None))
            return o

        return f

    N["java/lang/StringBuilder.<new>"] = sb_new
    N["java/lang/StringBuilder.<init>"] = sb_init
    for d, full in (("I", "(I)"), ("J", "(J)"), ("C", "(C)"), ("Z", "(Z)"), ("F", "(F)"), ("D", "(D)"), ("[C", "([C)"), ("[C", "([CII)"),
                    ("L", "(Ljava/lang/String;)"), ("L", "(Ljava/lang/Object;)"), ("L", "(Ljava/lang/CharSequence;)"),
                    ("L", "(Ljava/lang/CharSequence;II)"), ("L", "(Ljava/lang/StringBuffer;)")):
        N[f"java/lang/StringBuilder.append:{full}Ljava/lang/StringBuilder;"] = sb_append(d)
    N["java/lang/StringBuilder.toString"] = lambda j, o: "".join(o.native)
    N["java/lang/StringBuilder.length"] = lambda j, o: len(o.native)
    N["java/lang/StringBuilder.charAt"] = lambda j, o, i: ord(o.native[i]) if 0 <= i < len(o.native) else j.throw("java/lang/StringIndexOutOfBoundsException")
    N["java/lang/StringBuilder.reverse"] = lambda j, o: (o.native.reverse(), o)[1]
    N["java/lang/StringBuilder.setLength"] = lambda j, o, n: (o.native.__setitem__(slice(None), (o.native + ["\0"] * n)[:n]), None)[1]

    def sb_set_char(j, o, i, c):
        if not 0 <= i < len(o.native):
            j.throw("java/lang/StringIndexOutOfBoundsException")
        o.native[i] = chr(c)

    def sb_insert(j, o, i, v, d=None):
        if not 0 <= i <= len(o.native):
            j.throw("java/lang/StringIndexOutOfBoundsException")
        o.native[i:i] = list(j.to_jstring(v, d))
        return o

    def sb_delete_char(j, o, i):
        if not 0 <= i < len(o.native):
            j.throw("java/lang/StringIndexOutOfBoundsException")
        del o.native[i]
        return o

    N["java/lang/StringBuilder.setCharAt"] = sb_set_char
    N["java/lang/StringBuilder.insert:(ILjava/lang/String;)Ljava/lang/StringBuilder;"] = lambda j, o, i, v: sb_insert(j, o, i, v)
    N["java/lang/StringBuilder.insert:(IC)Ljava/lang/StringBuilder;"] = lambda j, o, i, v: sb_insert(j, o, i, v, "C")
    N["java/lang/StringBuilder.insert:(II)Ljava/lang/StringBuilder;"] = lambda j, o, i, v: sb_insert(j, o, i, v, "I")
    N["java/lang/StringBuilder.deleteCharAt"] = sb_delete_char
    N["java/lang/StringBuilder.substring"] = lambda j, o, b, e=None: "".join(o.native[b:e])
    N["java/lang/StringBuilder.indexOf:(Ljava/lang/String;)I"] = lambda j, o, t: "".join(o.native).find(t)

    # ---- java.lang.Math ------------------------------------------------------------------------------------------------
    N["java/lang/Math.max"] = lambda j, a, b: (a if a != a else b if b != b else max(a, b)) if isinstance(a, float) or isinstance(b, float) else max(a, b)
    N["java/lang/Math.min"] = lambda j, a, b: (a if a != a else b if b != b else min(a, b)) if isinstance(a, float) or isinstance(b, float) else min(a, b)
    N["java/lang/Math.abs:(I)I"] = lambda j, a: i32(abs(a))
    N["java/lang/Math.abs:(J)J"] = lambda j, a: i64(abs(a))
    N["java/lang/Math.abs:(F)F"] = lambda j, a: abs(a)
    N["java/lang/Math.abs:(D)D"] = lambda j, a: abs(a)
    N["java/lang/Math.round:(F)I"] = lambda j, a: java_round(a, True)
    N["java/lang/Math.round:(D)J"] = lambda j, a: java_round(a, False)
    N["java/lang/Math.floor"] = lambda j, a: float(math.floor(a)) if math.isfinite(a) else a
    N["java/lang/Math.ceil"] = lambda j, a: float(math.ceil(a)) if math.isfinite(a) else a
    N["java/lang/Math.sqrt"] = lambda j, a: math.sqrt(a) if a >= 0 else math.nan
    N["java/lang/Math.pow"] = lambda j, a, b: _pow(a, b)
    N["java/lang/Math.log"] = lambda j, a: math.log(a) if a > 0 else (-math.inf if a == 0 else math.nan)
    N["java/lang/Math.log10"] = lambda j, a: math.log10(a) if a > 0 else (-math.inf if a == 0 else math.nan)
    N["java/lang/Math.floorDiv:(II)I"] = lambda j, a, b: i32(a // b)
    N["java/lang/Math.floorMod:(II)I"] = lambda j, a, b: i32(a % b)

    def _pow(a, b):
        try:
            return math.pow(a, b)
        except OverflowError:
            return math.inf
        except ValueError:
            return math.nan

    # ---- boxing ------------------------------------------------------------------------------------------------------
    for cls, prim, conv in (("Integer", "I", i32), ("Long", "J", i64), ("Byte", "B", lambda v: v), ("Short", "S", lambda v: v),
                            ("Character", "C", lambda v: v), ("Boolean", "Z", lambda v: v), ("Float", "F", f32), ("Double", "D", float)):
        c = f"java/lang/{cls}"
        N[f"{c}.valueOf:({prim})L{c};"] = (lambda cc: lambda j, v: JBox(cc, v))(c)
        N[f"{c}.<init>:({prim})V"] = (lambda cc: lambda j, o, v: JBox(cc, v))(c)
        N[f"{c}.equals"] = lambda j, a, b: 1 if isinstance(b, JBox) and a == b else 0
        N[f"{c}.toString:()Ljava/lang/String;"] = lambda j, a: j.to_jstring(a)
        N[f"{c}.compareTo"] = lambda j, a, b: (a.v > b.v) - (a.v < b.v)
    for cls in ("Integer", "Long", "Byte", "Short", "Float", "Double", "Number"):
        c = f"java/lang/{cls}"
        N[f"{c}.intValue"] = lambda j, a: i32(int(a.v)) if not isinstance(a.v, float) else JVM_f2i(a.v, 32)
        N[f"{c}.longValue"] = lambda j, a: i64(int(a.v)) if not isinstance(a.v, float) else JVM_f2i(a.v, 64)
        N[f"{c}.floatValue"] = lambda j, a: f32(float(a.v))
        N[f"{c}.doubleValue"] = lambda j, a: float(a.v)
        N[f"{c}.byteValue"] = lambda j, a: ((int(a.v) + 128) & 255) - 128
        N[f"{c}.shortValue"] = lambda j, a: ((int(a.v) + 32768) & 65535) - 32768
    N["java/lang/Character.charValue"] = lambda j, a: a.v
    N["java/lang/Boolean.booleanValue"] = lambda j, a: a.v
    N["java/lang/Integer.hashCode:()I"] = lambda j, a: a.v
    N["java/lang/Long.hashCode:()I"] = lambda j, a: i32(a.v ^ ((a.v & 0xFFFFFFFFFFFFFFFF) >> 32))
    N["java/lang/Long.hashCode:(J)I"] = lambda j, v: i32(v ^ ((v & 0xFFFFFFFFFFFFFFFF) >> 32))
    N["java/lang/Integer.hashCode:(I)I"] = lambda j, v: v
    N["java/lang/Integer.compare"] = lambda j, a, b: (a > b) - (a < b)
    N["java/lang/Long.compare"] = lambda j, a, b: (a > b) - (a < b)
    N["java/lang/Float.compare"] = lambda j, a, b: _fcompare(a, b)
    N["java/lang/Double.compare"] = lambda j, a, b: _fcompare(a, b)
    N["java/lang/Integer.sum:(II)I"] = lambda j, a, b: i32(a + b)
    N["java/lang/Integer.max:(II)I"] = lambda j, a, b: max(a, b)
    N["java/lang/Integer.min:(II)I"] = lambda j, a, b: min(a, b)
    N["java/lang/Long.sum:(JJ)J"] = lambda j, a, b: i64(a + b)
    N["java/lang/Double.isNaN:(D)Z"] = lambda j, v: 1 if v != v else 0
    N["java/lang/Double.isInfinite:(D)Z"] = lambda j, v: 1 if v in (math.inf, -math.inf) else 0
    N["java/lang/Double.POSITIVE_INFINITY"] = lambda j: math.inf
    N["java/lang/Double.NEGATIVE_INFINITY"] = lambda j: -math.inf
    N["java/lang/Double.NaN"] = lambda j: math.nan
    N["java/lang/Double.MAX_VALUE"] = lambda j: 1.7976931348623157e308
    N["java/lang/Integer.toString:(I)Ljava/lang/String;"] = lambda j, v: str(v)
    N["java/lang/Long.toString:(J)Ljava/lang/String;"] = lambda j, v: str(v)
    N["java/lang/Float.toString:(F)Ljava/lang/String;"] = lambda j, v: float_to_string(v, True)
    N["java/lang/Double.toString:(D)Ljava/lang/String;"] = lambda j, v: float_to_string(v, False)
    def to_radix(j, v, radix):
        if not 2 <= radix <= 36:
            radix = 10
        digits = "0123456789abcdefghijklmnopqrstuvwxyz"
        n, out = abs(v), ""
        while True:
            out = digits[n % radix] + out
            n //= radix
            if n == 0:
                break
        return ("-" if v < 0 else "") + out

    N["java/lang/Integer.toString:(II)Ljava/lang/String;"] = to_radix
    N["java/lang/Long.toString:(JI)Ljava/lang/String;"] = to_radix
    N["java/lang/Integer.parseInt:(Ljava/lang/String;I)I"] = lambda j, s, r: int(s, r)
    N["java/lang/Integer.parseInt:(Ljava/lang/String;)I"] = lambda j, s: _parse_int(j, s, 32)
    N["java/lang/Long.parseLong:(Ljava/lang/String;)J"] = lambda j, s: _parse_int(j, s, 64)
    N["java/lang/Integer.valueOf:(Ljava/lang/String;)Ljava/lang/Integer;"] = lambda j, s: JBox("java/lang/Integer", _parse_int(j, s, 32))
    N["java/lang/Integer.<init>:(Ljava/lang/String;)V"] = lambda j, _u, s: JBox("java/lang/Integer", _parse_int(j, s, 32))   # new Integer("12")
    N["java/lang/Integer.<init>:(I)V"] = lambda j, _u, v: JBox("java/lang/Integer", v)
    def parse_float(j, t, single):
        u = t.strip()
        try:
            if not u or any(ch not in "0123456789+-.eE" for ch in u.rstrip("fFdD")):
                raise ValueError
            v = float(u.rstrip("fFdD"))
        except ValueError:
            j.throw("java/lang/NumberFormatException", f'For input string: "{t}"')
        return f32(v) if single else v      # decimal -> nearest double -> nearest float; exact for the short decimals in read names

    N["java/lang/Float.parseFloat:(Ljava/lang/String;)F"] = lambda j, t: parse_float(j, t, True)
    N["java/lang/Double.parseDouble:(Ljava/lang/String;)D"] = lambda j, t: parse_float(j, t, False)
    N["java/lang/Float.valueOf:(Ljava/lang/String;)Ljava/lang/Float;"] = lambda j, t: JBox("java/lang/Float", parse_float(j, t, True))
    N["java/lang/Integer.bitCount"] = lambda j, v: bin(v & 0xFFFFFFFF).count("1")
    N["java/lang/Long.bitCount"] = lambda j, v: bin(v & 0xFFFFFFFFFFFFFFFF).count("1")
    N["java/lang/Long.toBinaryString"] = lambda j, v: bin(v & 0xFFFFFFFFFFFFFFFF)[2:]
    N["java/lang/Integer.toBinaryString"] = lambda j, v: bin(v & 0xFFFFFFFF)[2:]
    N["java/lang/Long.toHexString"] = lambda j, v: hex(v & 0xFFFFFFFFFFFFFFFF)[2:]
    N["java/lang/Integer.MAX_VALUE"] = lambda j: 2 ** 31 - 1
    N["java/lang/Integer.MIN_VALUE"] = lambda j: -2 ** 31
    N["java/lang/Long.MAX_VALUE"] = lambda j: 2 ** 63 - 1
    N["java/lang/Float.MAX_VALUE"] = lambda j: f32(3.4028234663852886e38)
    N["java/lang/Boolean.toString:(Z)Ljava/lang/String;"] = lambda j, v: "true" if v else "false"
    N["java/lang/Boolean.TRUE"] = lambda j: JBox("java/lang/Boolean", 1)
    N["java/lang/Boolean.FALSE"] = lambda j: JBox("java/lang/Boolean", 0)
    N["java/lang/Character.toUpperCase:(C)C"] = lambda j, c: ord(chr(c).upper()) if c < 128 else c
    N["java/lang/Character.toLowerCase:(C)C"] = lambda j, c: ord(chr(c).lower()) if c < 128 else c
    N["java/lang/Character.isDigit:(C)Z"] = lambda j, c: 1 if 48 <= c <= 57 else 0

    def JVM_f2i(v, bits):
        from jvm_exec import JVM

        return JVM._f2int(v, bits)

    def _fcompare(a, b):
        if a != a or b != b:
            return (a != a) - (b != b)
        if a == b == 0:
            sa, sb_ = math.copysign(1, a), math.copysign(1, b)
            return (sa > sb_) - (sa < sb_)
        return (a > b) - (a < b)

    def _parse_int(j, s, bits):
        t = s
        ok = t and (t[0] in "+-" and t[1:].isdigit() and t[1:].isascii() or t.isdigit() and t.isascii())
        if not ok:
            j.throw("java/lang/NumberFormatException", f'For input string: "{s}"')
        v = int(t)
        if not -(1 << (bits - 1)) <= v < (1 << (bits - 1)):
            j.throw("java/lang/NumberFormatException", f'For input string: "{s}"')
        return v

    # ---- System / Arrays / Objects -------------------------------------------------------------------------------------
    def arraycopy(j, src, sp, dst, dp, n):
        if src is None or dst is None:
            _npe(j)
        if sp < 0 or dp < 0 or n < 0 or sp + n > len(src.a) or dp + n > len(dst.a):
            j.throw("java/lang/ArrayIndexOutOfBoundsException", "arraycopy")
        dst.a[dp:dp + n] = src.a[sp:sp + n]

    N["java/lang/System.arraycopy"] = arraycopy
    N["java/lang/System.out"] = lambda j: _stream(j)
    N["java/lang/System.err"] = lambda j: _stream(j)

    def _stream(j):
        o = JObject("java/io/PrintStream")
        o.native = []
        return o

    N["java/io/PrintStream.println"] = lambda j, o, *a: None
    N["java/io/PrintStream.print"] = lambda j, o, *a: None

    def arr_fill(j, a, *r):
        if len(r) == 1:
            a.a[:] = [r[0]] * len(a.a)
        else:
            a.a[r[0]:r[1]] = [r[2]] * (r[1] - r[0])

    def arr_copy_of(j, a, n):
        if n < 0:
            j.throw("java/lang/NegativeArraySizeException")
        from jvm_exec import default_value

        return JArray(a.etype, (a.a + [default_value(a.etype)] * max(0, n - len(a.a)))[:n])

    def arr_copy_range(j, a, b, e):
        if b < 0 or b > len(a.a):
            j.throw("java/lang/ArrayIndexOutOfBoundsException")
        if b > e:
            j.throw("java/lang/IllegalArgumentException")
        from jvm_exec import default_value

        return JArray(a.etype, (a.a[b:e] + [default_value(a.etype)] * max(0, e - len(a.a))))

    def arr_equals(j, a, b):
        if a is None or b is None:
            return 1 if a is b else 0
        return 1 if a.a == b.a else 0

    def arr_hash(j, a):
        if a is None:
            return 0
        h = 1
        for v in a.a:
            if a.etype == "J":
                e = i32(v ^ ((v & 0xFFFFFFFFFFFFFFFF) >> 32))
            elif a.etype in ("B", "I", "S", "C"):
                e = v
            else:
                raise Unsupported("Arrays.hashCode element type")
            h = (31 * h + e) & 0xFFFFFFFF
        return i32(h)

    N["java/util/Arrays.fill"] = arr_fill
    N["java/util/Arrays.copyOf"] = arr_copy_of
    N["java/util/Arrays.copyOfRange"] = arr_copy_range
    N["java/util/Arrays.equals"] = arr_equals
    N["java/util/Arrays.hashCode"] = arr_hash
    N["java/util/Objects.requireNonNull"] = lambda j, o, *a: o if o is not None else _npe(j)
    N["java/util/Objects.equals"] = lambda j, a, b: 1 if (a is b or (a is not None and j.call_virtual(a, "equals", "(Ljava/lang/Object;)Z", b))) else 0

    # clone() of arrays
    N["[.clone"] = lambda j, a: JArray(a.etype, list(a.a))

    # ---- enums (their own bytecode runs; Enum's constructor / ordinal / name are language-level) -------------------------
    def enum_init(j, o, name, ordinal):
        o.f["$name"], o.f["$ordinal"] = name, ordinal

    N["java/lang/Enum.<init>:(Ljava/lang/String;I)V"] = enum_init
    N["java/lang/Enum.ordinal"] = lambda j, o: o.f["$ordinal"]
    N["java/lang/Enum.name"] = lambda j, o: o.f["$name"]
    N["java/lang/Enum.toString"] = lambda j, o: o.f["$name"]
    N["java/lang/Enum.equals"] = lambda j, a, b: 1 if a is b else 0
    N["java/lang/Enum.hashCode"] = lambda j, o: o.f["$ordinal"]
    N["java/lang/Enum.compareTo"] = lambda j, a, b: a.f["$ordinal"] - b.f["$ordinal"]
    TIER_A.update({"java/lang/Enum.<init>:(Ljava/lang/String;I)V", "java/lang/Enum.ordinal"})

    # ---- exceptions: constructors only store the message ----------------------------------------------------------------
    def exc_init(j, o, *a):
        o.f["message"] = a[0] if a and isinstance(a[0], str) else None

    for c in ("java/lang/Throwable", "java/lang/Exception", "java/lang/RuntimeException", "java/lang/IllegalArgumentException",
              "java/lang/IllegalStateException", "java/lang/IndexOutOfBoundsException", "java/lang/ArrayIndexOutOfBoundsException",
              "java/lang/UnsupportedOperationException", "java/lang/NullPointerException", "java/lang/Error",
              "java/lang/NumberFormatException", "java/lang/CloneNotSupportedException"):
        N[f"{c}.<init>"] = exc_init
    N["java/lang/Throwable.getMessage"] = lambda j, o: o.f.get("message")
    N["java/lang/Throwable.toString"] = lambda j, o: o.cls.replace("/", ".") + (": " + o.f["message"] if o.f.get("message") else "")
    N["java/lang/Throwable.printStackTrace"] = lambda j, o, *a: None

    # ---- tier C: ordered containers as Python lists; membership-only sets ------------------------------------------------
    def list_new(cls):
        def f(j):
            o = JObject(cls)
            o.native = []
            return o

        return f

    def list_init(j, o, *a):
        if o.native is None:
            o.native = []   # a jar class extending ArrayList
        if a and isinstance(a[0], JObject) and isinstance(a[0].native, list):
            o.native = list(a[0].native)

    def _eq(j, a, b):
        if a is b:
            return True
        if a is None or b is None:
            return False
        if isinstance(a, (str, JBox)):
            return a == b
        return bool(j.call_virtual(a, "equals", "(Ljava/lang/Object;)Z", b))

    def list_get(j, o, i):
        if not 0 <= i < len(o.native):
            j.throw("java/lang/IndexOutOfBoundsException", f"Index {i} out of bounds for length {len(o.native)}")
        return o.native[i]

    def list_set(j, o, i, v):
        if not 0 <= i < len(o.native):
            j.throw("java/lang/IndexOutOfBoundsException")
        old = o.native[i]
        o.native[i] = v
        return old

    def list_remove_idx(j, o, i):
        if not 0 <= i < len(o.native):
            j.throw("java/lang/IndexOutOfBoundsException")
        return o.native.pop(i)

    def list_remove_obj(j, o, v):
        for k, e in enumerate(o.native):
            if _eq(j, v, e):
                del o.native[k]
                return 1
        return 0

    def no_such(j):
        j.throw("java/util/NoSuchElementException")

    for c in ("java/util/ArrayList", "java/util/LinkedList", "java/util/ArrayDeque"):
        N[f"{c}.<new>"] = list_new(c)
        N[f"{c}.<init>"] = list_init
        N[f"{c}.size"] = lambda j, o: len(o.native)
        N[f"{c}.isEmpty"] = lambda j, o: 1 if not o.native else 0
        N[f"{c}.clear"] = lambda j, o: o.native.clear()
        N[f"{c}.add:(Ljava/lang/Object;)Z"] = lambda j, o, v: (o.native.append(v), 1)[1]
        N[f"{c}.addLast"] = lambda j, o, v: o.native.append(v)
        N[f"{c}.offer"] = lambda j, o, v: (o.native.append(v), 1)[1]
        N[f"{c}.offerLast"] = lambda j, o, v: (o.native.append(v), 1)[1]
        N[f"{c}.addFirst"] = lambda j, o, v: o.native.insert(0, v)
        N[f"{c}.push"] = lambda j, o, v: o.native.insert(0, v)
        N[f"{c}.pop"] = lambda j, o: o.native.pop(0) if o.native else no_such(j)
        N[f"{c}.removeFirst"] = lambda j, o: o.native.pop(0) if o.native else no_such(j)
        N[f"{c}.removeLast"] = lambda j, o: o.native.pop() if o.native else no_such(j)
        N[f"{c}.pollFirst"] = lambda j, o: o.native.pop(0) if o.native else None
        N[f"{c}.poll"] = lambda j, o: o.native.pop(0) if o.native else None
        N[f"{c}.pollLast"] = lambda j, o: o.native.pop() if o.native else None
        N[f"{c}.peekFirst"] = lambda j, o: o.native[0] if o.native else None
        N[f"{c}.peek"] = lambda j, o: o.native[0] if o.native else None
        N[f"{c}.peekLast"] = lambda j, o: o.native[-1] if o.native else None
        N[f"{c}.getFirst"] = lambda j, o: o.native[0] if o.native else no_such(j)
        N[f"{c}.getLast"] = lambda j, o: o.native[-1] if o.native else no_such(j)
        N[f"{c}.contains"] = lambda j, o, v: 1 if any(_eq(j, v, e) for e in o.native) else 0
    def list_remove_all(j, o, coll):
        """AbstractCollection.removeAll: every element e of the list for which coll.contains(e) (coll: an ordered list here; contains = equals())"""
        if not isinstance(coll.native, list):
            raise Unsupported(f"removeAll({coll.cls}): only ordered collections")
        keep = [e for e in o.native if not any(_eq(j, e, x) for x in coll.native)]   # contains(e) tests o.equals(e) with o = e, the list element, as argument: x.equals(e)
        changed = len(keep) != len(o.native)
        o.native[:] = keep
        return 1 if changed else 0

    def to_array(j, o, a=None):
        if a is None:
            return JArray("Ljava/lang/Object;", list(o.native))
        if isinstance(a, JLambda):       # Collection.toArray(IntFunction) (a default method since Java 11): toArray(generator.apply(0))
            a = j.call_fn(j, a, 0)
        if len(a.a) < len(o.native):
            return JArray(a.etype, list(o.native))
        a.a[:len(o.native)] = o.native
        if len(a.a) > len(o.native):
            a.a[len(o.native)] = None
        return a

    for c in ("java/util/ArrayList", "java/util/LinkedList", "java/util/ArrayDeque"):
        N[f"{c}.toArray"] = to_array
    for c in ("java/util/ArrayList", "java/util/LinkedList"):
        N[f"{c}.get"] = list_get
        N[f"{c}.set"] = list_set
        N[f"{c}.add:(ILjava/lang/Object;)V"] = lambda j, o, i, v: o.native.insert(i, v)
        N[f"{c}.remove:(I)Ljava/lang/Object;"] = list_remove_idx
        N[f"{c}.remove:(Ljava/lang/Object;)Z"] = list_remove_obj
        N[f"{c}.addAll:(Ljava/util/Collection;)Z"] = lambda j, o, c2: (o.native.extend(c2.native), 1 if c2.native else 0)[1]
        N[f"{c}.indexOf"] = lambda j, o, v: next((k for k, e in enumerate(o.native) if _eq(j, v, e)), -1)
        N[f"{c}.removeAll"] = list_remove_all
    # interface views of the same objects
    for iface, impl in (("java/util/List", "java/util/ArrayList"), ("java/util/Collection", "java/util/ArrayList"),
                        ("java/util/Deque", "java/util/ArrayDeque"), ("java/util/Queue", "java/util/ArrayDeque")):
        for k in list(N):
            if k.startswith(impl + ".") and not k.endswith(("<new>", "<init>")):
                N.setdefault(iface + k[len(impl):], N[k])

    # iterators over ordered containers (for-each loops)
    def iterator(j, o):
        if not isinstance(o.native, list):
            raise Unsupported(f"iterator over {o.cls}: hash-ordered iteration is not emulated")
        it = JObject("$ListIterator")
        it.native = [o, 0]
        return it

    for c in ("java/util/ArrayList", "java/util/LinkedList", "java/util/ArrayDeque", "java/util/List", "java/util/Collection",
              "java/lang/Iterable", "java/util/Deque"):
        N[f"{c}.iterator"] = iterator
    N["$ListIterator.hasNext"] = lambda j, it: 1 if it.native[1] < len(it.native[0].native) else 0

    def it_next(j, it):
        o, k = it.native
        if k >= len(o.native):
            no_such(j)
        it.native[1] = k + 1
        return o.native[k]

    N["$ListIterator.next"] = it_next

    # membership-only sets (the fastutil / eclipse-collections jars are absent from the checkout): add / contains / size.
    # Their hashing is NOT emulated and no iteration is offered, so nothing order-dependent can come out of them.
    def set_new(cls):
        def f(j):
            o = JObject(cls)
            o.native = set()
            return o

        return f

    def set_add(j, o, v):
        k = v.v if isinstance(v, JBox) else v
        if k in o.native:
            return 0
        o.native.add(k)
        return 1

    for c in ("it/unimi/dsi/fastutil/longs/LongOpenHashSet", "it/unimi/dsi/fastutil/ints/IntOpenHashSet",
              "org/eclipse/collections/impl/set/mutable/primitive/IntHashSet", "org/eclipse/collections/impl/set/mutable/primitive/LongHashSet"):
        N[f"{c}.<new>"] = set_new(c)
        N[f"{c}.<init>"] = lambda j, o, *a: None
        N[f"{c}.add"] = set_add
        N[f"{c}.contains"] = lambda j, o, v: 1 if (v.v if isinstance(v, JBox) else v) in o.native else 0
        N[f"{c}.size"] = lambda j, o: len(o.native)
        N[f"{c}.isEmpty"] = lambda j, o: 1 if not o.native else 0
        N[f"{c}.clear"] = lambda j, o: o.native.clear()
    for iface, impl in (("it/unimi/dsi/fastutil/longs/LongSet", "it/unimi/dsi/fastutil/longs/LongOpenHashSet"),
                        ("it/unimi/dsi/fastutil/ints/IntSet", "it/unimi/dsi/fastutil/ints/IntOpenHashSet"),
                        ("org/eclipse/collections/api/set/primitive/MutableIntSet", "org/eclipse/collections/impl/set/mutable/primitive/IntHashSet"),
                        ("org/eclipse/collections/api/set/primitive/MutableLongSet", "org/eclipse/collections/impl/set/mutable/primitive/LongHashSet")):
        for k in list(N):
            if k.startswith(impl + ".") and not k.endswith(("<new>", "<init>")):
                N.setdefault(iface + k[len(impl):], N[k])


# =====================================================================================================================
# tier C (continued): java.util.Optional and ORDERED, SEQUENTIAL java.util.stream pipelines over ordered sources.
# A stream is evaluated eagerly, stage by stage, as a Python list in encounter order (sorted() is the stable sort the
# specification requires for ordered streams).  This equals the lazy JDK evaluation whenever the lambdas of the
# intermediate stages are free of side effects, which holds for every pipeline the fixtures execute; parallel streams
# raise Unsupported (their result order is not defined by the specification for forEach).
# =====================================================================================================================
def install_streams(jvm):
    import functools

    N = jvm.natives

    def call_fn(j, f, *args):
        """invoke a functional-interface object: a lambda, or an instance of a jar class implementing the interface"""
        if isinstance(f, JLambda):
            return j.call_lambda(f, list(args))
        if isinstance(f, JObject) and f.cls == "$Comparator":
            return f.native(*args)
        if isinstance(f, JObject) and f.cls == "$Fn":
            return f.native(*args)
        jc = j.load(j.class_of(f))
        cands = [m for (n, d), m in jc.methods.items() if m.code is not None and not m.static and len(m.args) == len(args)
                 and n in ("apply", "test", "accept", "compare", "applyAsInt", "applyAsLong", "applyAsDouble", "get", "call")
                 and not (m.acc & 0x1000 and m.acc & 0x0040)]
        if len(cands) != 1:
            raise Unsupported(f"functional object of class {f.cls}")
        return j.run(cands[0], [f] + list(args))

    jvm.call_fn = call_fn

    def truth(v):
        return bool(v.v) if isinstance(v, JBox) else bool(v)

    def unbox(v):
        return v.v if isinstance(v, JBox) else v

    def mk(kind, items):
        o = JObject({"ref": "java/util/stream/Stream", "int": "java/util/stream/IntStream", "long": "java/util/stream/LongStream",
                     "double": "java/util/stream/DoubleStream"}[kind])
        o.native = list(items)
        return o

    def kind_of(o):
        return {"java/util/stream/Stream": "ref", "java/util/stream/IntStream": "int", "java/util/stream/LongStream": "long",
                "java/util/stream/DoubleStream": "double"}[o.cls]

    def natural_cmp(j, a, b):
        if isinstance(a, JBox):
            return (a.v > b.v) - (a.v < b.v)
        if isinstance(a, str):
            return N["java/lang/String.compareTo:(Ljava/lang/String;)I"](j, a, b)
        return j.call_virtual(a, "compareTo", "(Ljava/lang/Object;)I", b)

    def comparator(fn):
        o = JObject("$Comparator")
        o.native = fn
        return o

    def fn_obj(fn):
        o = JObject("$Fn")
        o.native = fn
        return o

    def stream_of_collection(j, c):
        if isinstance(c.native, list):
            return mk("ref", c.native)
        if hasattr(c.native, "iter_keys"):
            return mk("ref", c.native.iter_keys())
        raise Unsupported(f"stream over {c.cls}: hash-ordered iteration is not emulated at this tier")

    for c in ("java/util/ArrayList", "java/util/LinkedList", "java/util/ArrayDeque", "java/util/List", "java/util/Collection",
              "java/util/Set", "java/util/HashSet", "java/util/Deque"):
        N[f"{c}.stream"] = stream_of_collection
        N[f"{c}.parallelStream"] = lambda j, c_: (_ for _ in ()).throw(Unsupported("parallel stream"))
    def set_all(j, a, f):
        for k in range(len(a.a)):
            v = call_fn(j, f, k)
            a.a[k] = unbox(v) if a.etype in ("I", "J", "D") else v

    N["java/util/Arrays.setAll"] = set_all
    N["java/util/Arrays.stream"] = lambda j, a, *r: mk({"I": "int", "J": "long", "D": "double"}.get(a.etype, "ref"), a.a if not r else a.a[r[0]:r[1]])
    N["java/util/Arrays.asList"] = lambda j, a: _list(j, a.a)
    N["java/util/stream/Stream.of:([Ljava/lang/Object;)Ljava/util/stream/Stream;"] = lambda j, a: mk("ref", a.a)
    N["java/util/stream/Stream.of:(Ljava/lang/Object;)Ljava/util/stream/Stream;"] = lambda j, a: mk("ref", [a])
    N["java/util/stream/Stream.empty"] = lambda j: mk("ref", [])
    N["java/util/stream/IntStream.range"] = lambda j, a, b: mk("int", range(a, b))
    N["java/util/stream/IntStream.rangeClosed"] = lambda j, a, b: mk("int", range(a, b + 1))
    N["java/util/stream/IntStream.of:([I)Ljava/util/stream/IntStream;"] = lambda j, a: mk("int", a.a)
    N["java/lang/String.chars"] = lambda j, s: mk("int", [ord(c) for c in s])
    N["java/lang/CharSequence.chars"] = lambda j, s: mk("int", [ord(c) for c in j.to_jstring(s)])

    def _list(j, items):
        o = JObject("java/util/ArrayList")
        o.native = list(items)
        return o

    BOX = {"int": "java/lang/Integer", "long": "java/lang/Long", "double": "java/lang/Double"}

    def sorted_(j, s, cmp=None):
        k = kind_of(s)
        if k != "ref":
            return mk(k, sorted(s.native))
        if cmp is None:
            key = functools.cmp_to_key(lambda a, b: natural_cmp(j, a, b))
        else:
            key = functools.cmp_to_key(lambda a, b: call_fn(j, cmp, a, b))
        return mk("ref", sorted(s.native, key=key))

    def parallel(j, s):
        raise Unsupported("parallel stream")

    def reduce_(j, s, *a):
        items = list(s.native)
        if len(a) == 2:
            acc = a[0]
            for v in items:
                acc = call_fn(j, a[1], acc, v)
            return acc
        if not items:
            return optional(j, None, kind_of(s))
        acc = items[0]
        for v in items[1:]:
            acc = call_fn(j, a[0], acc, v)
        return optional(j, acc, kind_of(s))

    def optional(j, v, kind="ref"):
        o = JObject({"ref": "java/util/Optional", "int": "java/util/OptionalInt", "long": "java/util/OptionalLong",
                     "double": "java/util/OptionalDouble"}[kind])
        o.native = (v,)
        return o

    def max_min(j, s, cmp, sign):
        items = list(s.native)
        if not items:
            return optional(j, None, kind_of(s))
        best = items[0]
        for v in items[1:]:
            c = call_fn(j, cmp, v, best) if kind_of(s) == "ref" else (v > best) - (v < best)
            # Stream.max = reduce(BinaryOperator.maxBy(cmp)): maxBy keeps the LEFT operand on a tie; minBy likewise
            if sign * c > 0:
                best = v
        return optional(j, best, kind_of(s))

    def collect(j, s, col):
        if not (isinstance(col, JObject) and col.cls == "$Collector"):
            raise Unsupported("collect with a user collector")
        return col.native(j, list(s.native))

    def flat_map(j, s, f):
        out = []
        for v in s.native:
            r = call_fn(j, f, v)
            if r is not None:
                out.extend(r.native)
        return mk("ref", out)

    def distinct(j, s):
        out = []
        for v in s.native:
            if not any(_eq(j, v, e) for e in out):
                out.append(v)
        return mk(kind_of(s), out)

    def _eq(j, a, b):
        if a is b:
            return True
        if a is None or b is None:
            return False
        if isinstance(a, (str, JBox, int, float)):
            return a == b
        return bool(j.call_virtual(a, "equals", "(Ljava/lang/Object;)Z", b))

    for c, k in (("java/util/stream/Stream", "ref"), ("java/util/stream/IntStream", "int"), ("java/util/stream/LongStream", "long"),
                 ("java/util/stream/DoubleStream", "double")):
        N[f"{c}.filter"] = lambda j, s, f: mk(kind_of(s), [v for v in s.native if truth(call_fn(j, f, v))])
        N[f"{c}.map"] = lambda j, s, f: mk(kind_of(s), [call_fn(j, f, v) for v in s.native])
        N[f"{c}.mapToInt"] = lambda j, s, f: mk("int", [unbox(call_fn(j, f, v)) for v in s.native])
        N[f"{c}.mapToLong"] = lambda j, s, f: mk("long", [unbox(call_fn(j, f, v)) for v in s.native])
        N[f"{c}.mapToDouble"] = lambda j, s, f: mk("double", [float(unbox(call_fn(j, f, v))) for v in s.native])
        N[f"{c}.mapToObj"] = lambda j, s, f: mk("ref", [call_fn(j, f, v) for v in s.native])
        N[f"{c}.boxed"] = lambda j, s: mk("ref", [JBox(BOX[kind_of(s)], v) for v in s.native])
        N[f"{c}.asLongStream"] = lambda j, s: mk("long", s.native)
        N[f"{c}.asDoubleStream"] = lambda j, s: mk("double", [float(v) for v in s.native])
        N[f"{c}.flatMap"] = flat_map
        N[f"{c}.distinct"] = distinct
        N[f"{c}.sorted:()L{c};"] = lambda j, s: sorted_(j, s)
        N[f"{c}.sorted:(Ljava/util/Comparator;)L{c};"] = lambda j, s, cmp: sorted_(j, s, cmp)
        N[f"{c}.limit"] = lambda j, s, n: mk(kind_of(s), s.native[:n])
        N[f"{c}.skip"] = lambda j, s, n: mk(kind_of(s), s.native[n:])
        N[f"{c}.sequential"] = lambda j, s: s
        N[f"{c}.parallel"] = parallel
        N[f"{c}.unordered"] = lambda j, s: s
        N[f"{c}.count"] = lambda j, s: len(s.native)
        N[f"{c}.sum"] = lambda j, s: (i32(sum(s.native)) if kind_of(s) == "int" else i64(sum(s.native)) if kind_of(s) == "long"
                                      else math.fsum(s.native) if False else _dsum(s.native))
        N[f"{c}.forEach"] = lambda j, s, f: [call_fn(j, f, v) for v in s.native] and None
        N[f"{c}.forEachOrdered"] = lambda j, s, f: [call_fn(j, f, v) for v in s.native] and None
        N[f"{c}.anyMatch"] = lambda j, s, f: 1 if any(truth(call_fn(j, f, v)) for v in s.native) else 0
        N[f"{c}.allMatch"] = lambda j, s, f: 1 if all(truth(call_fn(j, f, v)) for v in s.native) else 0
        N[f"{c}.noneMatch"] = lambda j, s, f: 0 if any(truth(call_fn(j, f, v)) for v in s.native) else 1
        N[f"{c}.findFirst"] = lambda j, s: optional(j, s.native[0] if s.native else None, kind_of(s))
        N[f"{c}.findAny"] = lambda j, s: find_any(j, s)
        N[f"{c}.reduce"] = reduce_
        N[f"{c}.collect"] = collect
        N[f"{c}.toArray"] = lambda j, s, *a: stream_to_array(j, s, *a)
        N[f"{c}.iterator"] = lambda j, s: N["java/util/ArrayList.iterator"](j, _list(j, s.native))
    N["java/util/stream/Stream.max"] = lambda j, s, cmp: max_min(j, s, cmp, 1)
    N["java/util/stream/Stream.min"] = lambda j, s, cmp: max_min(j, s, cmp, -1)
    for c in ("java/util/stream/IntStream", "java/util/stream/LongStream", "java/util/stream/DoubleStream"):
        N[f"{c}.max"] = lambda j, s: max_min(j, s, None, 1)
        N[f"{c}.min"] = lambda j, s: max_min(j, s, None, -1)
        N[f"{c}.average"] = lambda j, s: optional(j, (_dsum([float(v) for v in s.native]) / len(s.native)) if s.native else None, "double")

    def _dsum(vals):
        # DoubleStream.sum may compensate (Kahan); the fixtures only sum small integers held in doubles, where every
        # summation order and compensation gives the same exact result -- anything else is refused
        t = 0.0
        for v in vals:
            if v != int(v) or abs(v) > 2 ** 40:
                raise Unsupported("DoubleStream.sum over non-integral values (summation order unspecified)")
            t += v
        return t

    def stream_to_array(j, s, *a):
        if a:  # toArray(IntFunction<A[]> generator): the generator makes the (typed) array
            arr = call_fn(j, a[0], len(s.native))
            arr.a[:] = list(s.native)
            return arr
        return JArray({"int": "I", "long": "J", "double": "D"}.get(kind_of(s), "Ljava/lang/Object;"), list(s.native))

    def find_any(j, s):
        """which element is open in the specification; the drivers only use sections whose caller asks isPresent() -- the note travels in
        the fixture's native list so that a reader can check that"""
        j.natives_used.add("$findAny(first element; valid only where the caller asks isPresent)")
        return optional(j, s.native[0] if s.native else None, kind_of(s))

    # ---- Collectors (ordered results only)
    def collector(fn):
        o = JObject("$Collector")
        o.native = fn
        return o

    N["java/util/stream/Collectors.toList"] = lambda j: collector(lambda j_, items: _list(j_, items))
    N["java/util/stream/Collectors.counting"] = lambda j: collector(lambda j_, items: JBox("java/lang/Long", len(items)))
    N["java/util/stream/Collectors.joining:()Ljava/util/stream/Collector;"] = lambda j: collector(lambda j_, items: "".join(j_.to_jstring(v) for v in items))
    N["java/util/stream/Collectors.joining:(Ljava/lang/CharSequence;)Ljava/util/stream/Collector;"] = \
        lambda j, sep: collector(lambda j_, items: j_.to_jstring(sep).join(j_.to_jstring(v) for v in items))

    N["java/util/stream/Collectors.joining:(Ljava/lang/CharSequence;Ljava/lang/CharSequence;Ljava/lang/CharSequence;)Ljava/util/stream/Collector;"] = \
        lambda j, sep, pre, suf: collector(lambda j_, items: j_.to_jstring(pre) + j_.to_jstring(sep).join(j_.to_jstring(v) for v in items) + j_.to_jstring(suf))
    N["java/util/stream/Collectors.summingInt"] = \
        lambda j, f: collector(lambda j_, items: JBox("java/lang/Integer", i32(sum(call_fn(j_, f, v) for v in items))))

    def grouping_by(j, keyf, *rest):
        """groupingBy(classifier[, downstream]) -> java.util.HashMap: lists in encounter order; the map's own iteration order is
        as for every hash container here (varied, not emulated)"""
        down = rest[-1] if rest else None
        factory = rest[0] if len(rest) == 2 else None   # groupingBy(classifier, mapFactory, downstream)

        def run(j_, items):
            m = call_fn(j_, factory) if factory is not None else j_.natives["java/util/HashMap.<new>"](j_)
            for v in items:
                k = call_fn(j_, keyf, v)
                cell = m.native.find(k)
                if cell is None:
                    m.native.put(k, [v])
                else:
                    cell[1].append(v)
            for cell in m.native.order:
                cell[1] = _list(j_, cell[1]) if down is None else down.native(j_, cell[1])
            return m

        return collector(run)

    def mapping(j, f, down):
        return collector(lambda j_, items: down.native(j_, [call_fn(j_, f, v) for v in items]))

    def to_map(j, keyf, valf, *rest):
        """toMap(key, value[, merge[, mapFactory]]): accumulates with Map.merge in encounter order; a null value is a NullPointerException
        (HashMap.merge / Objects.requireNonNull), a duplicate key without a merge function an IllegalStateException"""
        merge = rest[0] if rest else None
        factory = rest[1] if len(rest) > 1 else None

        def run(j_, items):
            m = call_fn(j_, factory) if factory is not None else j_.natives["java/util/HashMap.<new>"](j_)
            for v in items:
                k, val = call_fn(j_, keyf, v), call_fn(j_, valf, v)
                if val is None:
                    j_.throw("java/lang/NullPointerException")
                cell = m.native.find(k)
                if cell is None or cell[1] is None:
                    m.native.put(k, val)
                elif merge is None:
                    j_.throw("java/lang/IllegalStateException", "Duplicate key")
                else:
                    nv = call_fn(j_, merge, cell[1], val)
                    if nv is None:
                        m.native.remove(k)
                    else:
                        cell[1] = nv
            return m

        return collector(run)

    def to_collection(j, sup):
        def run(j_, items):
            c = call_fn(j_, sup)
            for v in items:
                j_.call_virtual(c, "add", "(Ljava/lang/Object;)Z", v)
            return c

        return collector(run)

    N["java/util/stream/Collectors.toCollection"] = to_collection

    def to_set(j):
        def run(j_, items):
            c = j_.natives["java/util/HashSet.<new>"](j_)
            for v in items:
                c.native.put(v, True)
            return c

        return collector(run)

    N["java/util/stream/Collectors.toSet"] = to_set
    N["java/util/stream/Collectors.toMap"] = to_map
    N["java/util/stream/Collectors.groupingBy"] = grouping_by
    N["java/util/stream/Collectors.mapping"] = mapping
    N["java/util/stream/Collectors.toUnmodifiableList"] = N["java/util/stream/Collectors.toList"]

    # ---- Optional
    for c, k in (("java/util/Optional", "ref"), ("java/util/OptionalInt", "int"), ("java/util/OptionalLong", "long"), ("java/util/OptionalDouble", "double")):
        N[f"{c}.isPresent"] = lambda j, o: 1 if o.native[0] is not None else 0
        N[f"{c}.isEmpty"] = lambda j, o: 1 if o.native[0] is None else 0
        for g in ("get", "getAsInt", "getAsLong", "getAsDouble", "orElseThrow"):
            N[f"{c}.{g}"] = lambda j, o, *a: o.native[0] if o.native[0] is not None else j.throw("java/util/NoSuchElementException", "No value present")
        N[f"{c}.orElse"] = lambda j, o, d: o.native[0] if o.native[0] is not None else d
        N[f"{c}.ifPresent"] = lambda j, o, f: call_fn(j, f, o.native[0]) if o.native[0] is not None else None
    N["java/util/Optional.of"] = lambda j, v: optional(j, v) if v is not None else _npe(j)
    N["java/util/Optional.ofNullable"] = lambda j, v: optional(j, v)
    N["java/util/Optional.empty"] = lambda j: optional(j, None)
    N["java/util/Optional.map"] = lambda j, o, f: optional(j, call_fn(j, f, o.native[0])) if o.native[0] is not None else o
    N["java/util/Optional.equals"] = lambda j, a, b: 1 if isinstance(b, JObject) and b.cls == a.cls and _eq(j, a.native[0], b.native[0]) else 0

    # ---- Comparator factories
    def comparing(j, keyf, kind="ref"):
        return comparator(lambda a, b: natural_cmp(j, call_fn(j, keyf, a), call_fn(j, keyf, b)) if kind == "ref"
                          else _cmpnum(unbox(call_fn(j, keyf, a)), unbox(call_fn(j, keyf, b))))

    def _cmpnum(a, b):
        return (a > b) - (a < b)

    N["java/util/Comparator.comparing:(Ljava/util/function/Function;)Ljava/util/Comparator;"] = lambda j, f: comparing(j, f)
    N["java/util/Comparator.comparingInt"] = lambda j, f: comparing(j, f, "num")
    N["java/util/Comparator.comparingLong"] = lambda j, f: comparing(j, f, "num")
    N["java/util/Comparator.comparingDouble"] = lambda j, f: comparing(j, f, "num")
    N["java/util/Comparator.naturalOrder"] = lambda j: comparator(lambda a, b: natural_cmp(j, a, b))
    N["java/util/Comparator.reverseOrder"] = lambda j: comparator(lambda a, b: natural_cmp(j, b, a))
    N["java/util/Collections.reverseOrder:()Ljava/util/Comparator;"] = lambda j: comparator(lambda a, b: natural_cmp(j, b, a))
    N["java/util/Comparator.reversed"] = lambda j, c: comparator(lambda a, b: call_fn(j, c, b, a))
    N["$Comparator.reversed"] = N["java/util/Comparator.reversed"]
    N["$Comparator.compare"] = lambda j, c, a, b: c.native(a, b)

    def then_comparing(j, c, nxt, kind=None):
        second = nxt if kind is None else comparing(j, nxt, kind)

        def f(a, b):
            r = call_fn(j, c, a, b)
            return r if r != 0 else call_fn(j, second, a, b)

        return comparator(f)

    N["java/util/Comparator.thenComparing:(Ljava/util/Comparator;)Ljava/util/Comparator;"] = lambda j, c, n: then_comparing(j, c, n)
    N["java/util/Comparator.thenComparing:(Ljava/util/function/Function;)Ljava/util/Comparator;"] = lambda j, c, n: then_comparing(j, c, n, "ref")
    N["java/util/Comparator.thenComparingInt"] = lambda j, c, n: then_comparing(j, c, n, "num")
    for k in [k for k in N if k.startswith("java/util/Comparator.") and "thenComparing" in k]:
        N["$Comparator." + k.split(".", 1)[1]] = N[k]
    N["java/util/Collections.singleton"] = lambda j, v: _list(j, [v])          # one element: no order to speak of
    N["java/util/Collections.singletonList"] = lambda j, v: _list(j, [v])
    N["java/util/Collections.emptyList"] = lambda j: _list(j, [])
    N["java/util/Collections.emptySet"] = lambda j: _list(j, [])
    N["java/util/Collections.unmodifiableList"] = lambda j, c: c
    N["java/util/Collections.unmodifiableMap"] = lambda j, m: m
    N["java/util/Collections.unmodifiableSet"] = lambda j, m: m
    N["java/util/function/Function.identity"] = lambda j: fn_obj(lambda v: v)

    def list_sort(j, o, cmp):
        key = functools.cmp_to_key((lambda a, b: natural_cmp(j, a, b)) if cmp is None else (lambda a, b: call_fn(j, cmp, a, b)))
        o.native.sort(key=key)

    N["java/util/ArrayList.sort"] = list_sort
    N["java/util/List.sort"] = list_sort
    N["java/util/Collections.sort:(Ljava/util/List;)V"] = lambda j, o: list_sort(j, o, None)
    N["java/util/Collections.sort:(Ljava/util/List;Ljava/util/Comparator;)V"] = list_sort
    N["java/util/ArrayList.forEach"] = lambda j, o, f: [call_fn(j, f, v) for v in list(o.native)] and None
    N["java/util/List.forEach"] = N["java/util/ArrayList.forEach"]
    N["java/lang/Iterable.forEach"] = N["java/util/ArrayList.forEach"]
    N["java/util/ArrayList.removeIf"] = lambda j, o, f: _remove_if(j, o, f)
    N["java/util/List.removeIf"] = N["java/util/ArrayList.removeIf"]
    N["java/util/Collection.removeIf"] = N["java/util/ArrayList.removeIf"]

    def _remove_if(j, o, f):
        keep = [v for v in o.native if not truth(call_fn(j, f, v))]
        changed = len(keep) != len(o.native)
        o.native[:] = keep
        return 1 if changed else 0


# =====================================================================================================================
# tier C (continued): java.util.HashSet / HashMap as MEMBERSHIP structures.  Keys are compared through the key's own
# hashCode() / equals() -- the reference's bytecode for its own classes -- which is all the Set / Map contract needs for
# add / contains / get / put / remove / size.  Iteration from bytecode is refused (`Unsupported`): its order is an
# implementation detail of the JDK, not of the reference.  Drivers read the contents through `.native.items_in_insertion_order()`
# and must compare them as SETS.
# =====================================================================================================================
class HashStore:
    def __init__(self, jvm):
        self.j = jvm
        self.buckets = {}   # hashCode -> [[key, value], ...]
        self.order = []     # [key, value] cells in insertion order (driver-side reading only)
        jvm.store_serial = getattr(jvm, "store_serial", 0) + 1
        self.serial = jvm.store_serial   # shuffled orders differ from container to container (a chain of two iterations must not cancel)
        self.cap0 = 16      # table size the first put allocates (jdk order only)
        self.peak = 0       # largest size reached: the table never shrinks
        self.identity = False

    def _hash(self, k):
        j = self.j
        if k is None:
            return 0
        if isinstance(k, str):
            return jhash_str(k)
        if isinstance(k, JBox):
            c = k.cls.split("/")[-1]
            if c == "Long":
                return i32(k.v ^ ((k.v & 0xFFFFFFFFFFFFFFFF) >> 32))
            if c in ("Integer", "Short", "Byte", "Character"):
                return k.v
            if c == "Boolean":
                return 1231 if k.v else 1237
            if c == "Float":
                import struct

                return i32(struct.unpack("<I", struct.pack("<f", k.v))[0]) if k.v == k.v else 0x7fc00000
            raise Unsupported("hashCode of " + k.cls)
        t = j.find_method(j.class_of(k), "hashCode", "()I")
        if t is None or isinstance(t, str) and t.startswith("java/lang/Object"):
            self.identity = True
            return id(k) & 0x7FFFFFFF  # identity hash: membership by identity
        return j.invoke(t, [k])

    def _equal(self, a, b):
        j = self.j
        if a is b:
            return True
        if a is None or b is None:
            return False
        if isinstance(a, (str, JBox)):
            return a == b
        t = j.find_method(j.class_of(a), "equals", "(Ljava/lang/Object;)Z")
        if t is None or isinstance(t, str):
            return False
        return bool(j.invoke(t, [a, b]))

    def find(self, k):
        for cell in self.buckets.get(self._hash(k), ()):
            if self._equal(k, cell[0]):  # HashMap.getNode: key.equals(k) with the probe as receiver
                return cell
        return None

    def put(self, k, v):
        cell = self.find(k)
        if cell is not None:
            old = cell[1]
            cell[1] = v
            return old, False
        cell = [k, v]
        self.buckets.setdefault(self._hash(k), []).append(cell)
        self.order.append(cell)
        self.peak = max(self.peak, len(self.order))
        return None, True

    def remove(self, k):
        h = self._hash(k)
        for cell in self.buckets.get(h, ()):
            if self._equal(k, cell[0]):
                self.buckets[h].remove(cell)
                self.order.remove(cell)
                return cell
        return None

    def __len__(self):
        return len(self.order)

    def items_in_insertion_order(self):
        return [(c[0], c[1]) for c in self.order]

    def cells_for_iteration(self, what, cls):
        """Iteration order of a hash container is NOT emulated.  jvm.hash_order = None refuses it; otherwise it is one of
        'insertion' | 'reverse' | ('shuffle', seed): the fixture generator runs every case under several of these and keeps a
        case only if all of them give the same answer, i.e. only results that do not depend on the JDK's order."""
        mode = self.j.hash_order
        if mode is None:
            raise Unsupported(f"{what} of {cls}: hash-ordered iteration is not emulated (set jvm.hash_order to vary it)")
        if mode == "jdk":
            return self.cells_in_jdk_order(what, cls)
        self.j.natives_used.add("$hash-iteration(order varied, not emulated)")
        cells = list(self.order)
        if mode == "reverse":
            cells.reverse()
        elif isinstance(mode, tuple):
            import random

            random.Random(mode[1] * 1000003 + len(cells) + 7919 * self.serial).shuffle(cells)
        return cells


def _cells_in_jdk_order(self, what, cls):
    """Tier D, opt-in (jvm.hash_order = 'jdk'): the order java.util.HashMap (JDK 8 .. 21, the layout is part of its documented
    implementation notes) walks its table in -- buckets ascending, bucket of a key = (h ^ h >>> 16) & (capacity - 1) with h the key's own
    hashCode() as the bytecode / the value class defines it, capacity = the initial table size doubled while size exceeded 0.75 capacity,
    nodes of a bucket in the order they were linked (insertion order; resize() splits keep it, remove() unlinks).  Refused for keys with
    identity hashes and for bins long enough to be turned into trees (8 nodes at capacity >= 64), whose order this does not model."""
    if self.identity:
        raise Unsupported(f"{what} of {cls}: identity-hashed keys have no reproducible order")
    self.j.natives_used.add("$hash-iteration(jdk table order from the keys' hashCode())")
    cap = self.cap0
    while self.peak > (cap * 3) // 4:
        cap *= 2
    keyed = []
    per = {}
    for cell in self.order:
        h = self._hash(cell[0]) & 0xFFFFFFFF
        b = (h ^ (h >> 16)) & (cap - 1)
        per[b] = per.get(b, 0) + 1
        keyed.append((b, cell))
    if per and max(per.values()) >= 8:
        raise Unsupported(f"{what} of {cls}: a bin of {max(per.values())} nodes may have been treeified")
    keyed.sort(key=lambda t: t[0])
    return [c for _, c in keyed]


HashStore.cells_in_jdk_order = _cells_in_jdk_order


def _table_size_for(n):
    c = 1
    while c < n:
        c *= 2
    return max(c, 1)


def install_hash(jvm):
    N = jvm.natives

    def new(cls):
        def f(j):
            o = JObject(cls)
            o.native = HashStore(j)
            return o

        return f

    def init(j, o, *a):
        if o.native is None:
            o.native = HashStore(j)  # a jar class extending HashSet / HashMap
        if a and isinstance(a[0], int):
            o.native.cap0 = _table_size_for(a[0])  # HashMap(int): threshold = tableSizeFor(initialCapacity)
        if a and isinstance(a[0], JObject) and isinstance(a[0].native, list):
            if o.cls.endswith("Set"):
                o.native.cap0 = _table_size_for(max(int(f32(len(a[0].native) / f32(0.75))) + 1, 16))  # HashSet(Collection)
            for v in a[0].native:
                o.native.put(v, True)
        elif a and isinstance(a[0], JObject) and isinstance(a[0].native, HashStore):
            src = a[0].native
            cells = src.cells_for_iteration("copy constructor", o.cls) if j.hash_order == "jdk" else [tuple(c) for c in src.order]
            if o.cls.endswith("Set"):
                o.native.cap0 = _table_size_for(max(int(f32(len(cells) / f32(0.75))) + 1, 16))
            for k, v in cells:
                o.native.put(k, v)

    def refuse(what):
        def f(j, o, *a):
            raise Unsupported(f"{what} of {o.cls}: hash-ordered iteration is not emulated (tier C is membership only)")

        return f

    for c in ("java/util/HashSet", "java/util/HashMap"):
        N[f"{c}.<new>"] = new(c)
        N[f"{c}.<init>"] = init
        N[f"{c}.size"] = lambda j, o: len(o.native)
        N[f"{c}.isEmpty"] = lambda j, o: 0 if len(o.native) else 1
        N[f"{c}.clear"] = lambda j, o: o.native.__init__(j)
        for it in ("parallelStream", "toString", "hashCode"):
            N[f"{c}.{it}"] = refuse(it)

    def set_hash(j, o):
        """AbstractSet.hashCode: the sum of the members' hash codes -- independent of any order"""
        t = 0
        for cell in o.native.order:
            t = (t + o.native._hash(cell[0])) & 0xFFFFFFFF
        return i32(t)

    def set_equals(j, o, other):
        if o is other:
            return 1
        if not isinstance(other, JObject) or not isinstance(other.native, HashStore) or len(other.native) != len(o.native):
            return 0
        return 1 if all(other.native.find(cell[0]) is not None for cell in o.native.order) else 0

    N["java/util/HashSet.hashCode"] = set_hash
    N["java/util/HashSet.equals"] = set_equals

    def _aslist(j, items):
        o = JObject("java/util/ArrayList")
        o.native = list(items)
        return o

    def entry(k, v):
        e = JObject("java/util/AbstractMap$SimpleEntry")
        e.native = [k, v]
        return e

    def keys_of(j, o, what):
        return [c[0] for c in o.native.cells_for_iteration(what, o.cls)]

    N["$Entry.getKey"] = lambda j, e: e.native[0]
    N["$Entry.getValue"] = lambda j, e: e.native[1]
    N["java/util/Map$Entry.getKey"] = N["$Entry.getKey"]
    N["java/util/Map$Entry.getValue"] = N["$Entry.getValue"]
    N["java/util/AbstractMap$SimpleEntry.<new>"] = lambda j: entry(None, None)
    N["java/util/AbstractMap$SimpleEntry.<init>"] = lambda j, e, k, v: e.native.__setitem__(slice(None), [k, v])
    N["java/util/AbstractMap$SimpleEntry.getKey"] = N["$Entry.getKey"]
    N["java/util/AbstractMap$SimpleEntry.getValue"] = N["$Entry.getValue"]
    N["java/util/HashSet.iterator"] = lambda j, o: j.natives["java/util/ArrayList.iterator"](j, _aslist(j, keys_of(j, o, "iterator")))
    N["java/util/HashSet.stream"] = lambda j, o: j.natives["java/util/ArrayList.stream"](j, _aslist(j, keys_of(j, o, "stream")))
    N["java/util/HashSet.forEach"] = lambda j, o, f: [j.call_fn(j, f, k) for k in keys_of(j, o, "forEach")] and None
    N["java/util/HashSet.toArray"] = lambda j, o, *a: j.natives["java/util/ArrayList.toArray"](j, _aslist(j, keys_of(j, o, "toArray")), *a)
    N["java/util/HashSet.addAll"] = lambda j, o, c2: 1 if [o.native.put(v, True) for v in (c2.native if isinstance(c2.native, list) else keys_of(j, c2, "addAll"))] and False else 1
    N["java/util/HashMap.keySet"] = lambda j, o: _aslist(j, keys_of(j, o, "keySet"))
    N["java/util/HashMap.values"] = lambda j, o: _aslist(j, [c[1] for c in o.native.cells_for_iteration("values", o.cls)])
    N["java/util/HashMap.entrySet"] = lambda j, o: _aslist(j, [entry(c[0], c[1]) for c in o.native.cells_for_iteration("entrySet", o.cls)])
    N["java/util/HashMap.forEach"] = lambda j, o, f: [j.call_fn(j, f, c[0], c[1]) for c in o.native.cells_for_iteration("forEach", o.cls)] and None

    def compute_if_absent(j, o, k, f):
        c = o.native.find(k)
        if c is not None and c[1] is not None:
            return c[1]
        v = j.call_fn(j, f, k)
        if v is not None:
            o.native.put(k, v)
        return v

    N["java/util/HashMap.computeIfAbsent"] = compute_if_absent
    N["java/util/HashSet.add"] = lambda j, o, v: 1 if o.native.put(v, True)[1] else 0
    N["java/util/HashSet.contains"] = lambda j, o, v: 1 if o.native.find(v) is not None else 0
    N["java/util/HashSet.remove"] = lambda j, o, v: 1 if o.native.remove(v) is not None else 0
    N["java/util/HashMap.put"] = lambda j, o, k, v: o.native.put(k, v)[0]
    N["java/util/HashMap.get"] = lambda j, o, k: (o.native.find(k) or [None, None])[1]
    N["java/util/HashMap.containsKey"] = lambda j, o, k: 1 if o.native.find(k) is not None else 0
    N["java/util/HashMap.remove:(Ljava/lang/Object;)Ljava/lang/Object;"] = lambda j, o, k: (o.native.remove(k) or [None, None])[1]
    N["java/util/HashMap.getOrDefault"] = lambda j, o, k, d: (o.native.find(k) or [None, d])[1]

    def put_if_absent(j, o, k, v):
        c = o.native.find(k)
        if c is not None and c[1] is not None:
            return c[1]
        o.native.put(k, v)
        return None

    N["java/util/HashMap.putIfAbsent"] = put_if_absent

    # java.util.LinkedHashMap: the same keyed store, iterated in insertion order (specified, tier C)
    class LinkedStore(HashStore):
        def cells_for_iteration(self, what, cls):
            return list(self.order)

    def linked_new(j):
        o = JObject("java/util/LinkedHashMap")
        o.native = LinkedStore(j)
        return o

    N["java/util/LinkedHashMap.<new>"] = linked_new
    N["java/util/LinkedHashMap.<init>"] = lambda j, o, *a: None
    # java.util.EnumMap: the same keyed store, iterated in the order of the enum constants (specified: "natural order of its keys", tier C)
    class EnumStore(HashStore):
        def cells_for_iteration(self, what, cls):
            return sorted(self.order, key=lambda c: c[0].f["$ordinal"])

    def enum_map_new(j):
        o = JObject("java/util/EnumMap")
        o.native = EnumStore(j)
        return o

    N["java/util/EnumMap.<new>"] = enum_map_new
    N["java/util/EnumMap.<init>"] = lambda j, o, *a: None
    for k in list(N):
        if k.startswith("java/util/HashMap.") and not k.endswith(("<new>", "<init>")):
            N["java/util/EnumMap" + k[len("java/util/HashMap"):]] = N[k]
    # java.util.TreeMap: the same keyed store; get(null) throws as a TreeMap with natural ordering does
    # (iteration: ascending natural order of boxed-number or String keys, which is all the reference puts into one -- specified order, tier C)
    class TreeStore(HashStore):
        def cells_for_iteration(self, what, cls):
            def key(c):
                k = c[0]
                if isinstance(k, JBox):
                    return k.v
                if isinstance(k, str):
                    return k
                raise Unsupported(f"{what} of a TreeMap with keys of {getattr(k, 'cls', type(k))}")
            return sorted(self.order, key=key)

    def tree_new(j):
        o = JObject("java/util/TreeMap")
        o.native = TreeStore(j)
        return o

    N["java/util/TreeMap.<new>"] = tree_new
    N["java/util/TreeMap.<init>:()V"] = lambda j, o: None
    for k in ("entrySet", "keySet", "values", "forEach", "containsKey", "getOrDefault", "putIfAbsent", "computeIfAbsent", "isEmpty"):
        if "java/util/HashMap." + k in N:
            N["java/util/TreeMap." + k] = N["java/util/HashMap." + k]

    def tree_get(j, o, k):
        if k is None:
            j.throw("java/lang/NullPointerException")
        return (o.native.find(k) or [None, None])[1]

    N["java/util/TreeMap.get"] = tree_get
    N["java/util/TreeMap.put"] = lambda j, o, k, v: o.native.put(k, v)[0]
    N["java/util/TreeMap.size"] = lambda j, o: len(o.native)
    # ConcurrentHashMap used single-threaded: the same membership structure (its iteration order is likewise not emulated)
    for k in list(N):
        if k.startswith("java/util/HashMap."):
            N["java/util/concurrent/ConcurrentHashMap" + k[len("java/util/HashMap"):]] = N[k]
    N["java/util/concurrent/ConcurrentHashMap.<new>"] = new("java/util/concurrent/ConcurrentHashMap")
    N["java/util/concurrent/ConcurrentHashMap.newKeySet"] = lambda j, *a: new("java/util/HashSet")(j)
    for iface, impl in (("java/util/Set", "java/util/HashSet"), ("java/util/Map", "java/util/HashMap")):
        for k in list(N):
            if k.startswith(impl + ".") and not k.endswith(("<new>", "<init>")):
                N.setdefault(iface + k[len(impl):], N[k])
    jdk_super = __import__("jvm_exec").JDK_SUPER
    jdk_ifaces = __import__("jvm_exec").JDK_IFACES
    jdk_super.update({"java/util/HashSet": "java/util/AbstractSet", "java/util/AbstractSet": "java/util/AbstractCollection",
                      "java/util/HashMap": "java/util/AbstractMap", "java/util/AbstractMap": "java/lang/Object"})
    jdk_ifaces["java/util/AbstractMap$SimpleEntry"] = ["java/util/Map$Entry"]
    jdk_super["java/util/LinkedHashMap"] = "java/util/HashMap"
    jdk_super["java/util/TreeMap"] = "java/util/AbstractMap"
    jdk_ifaces["java/util/TreeMap"] = ["java/util/NavigableMap", "java/util/SortedMap", "java/util/Map"]
    jdk_super["java/util/concurrent/ConcurrentHashMap"] = "java/util/AbstractMap"
    jdk_ifaces["java/util/concurrent/ConcurrentHashMap"] = ["java/util/Map", "java/util/concurrent/ConcurrentMap"]
    jdk_ifaces.update({"java/util/HashSet": ["java/util/Set", "java/util/Collection", "java/lang/Iterable"], "java/util/HashMap": ["java/util/Map"]})


# =====================================================================================================================
# environment: things a driver has to answer for the JVM process (no bearing on the algorithms): runtime, clocks,
# properties, atomics used as plain counters, DecimalFormat for the two patterns the reference uses
# =====================================================================================================================
def install_treeset(jvm):
    """java.util.TreeSet (tier C: an ordered container): elements kept sorted by the comparator (or compareTo); an element that compares
    equal to a member is not added -- the specified behaviour of a sorted set, independent of its tree"""
    N = jvm.natives

    def cmp(j, o, a, b):
        c = o.native["cmp"]
        if c is None:
            if isinstance(a, JBox):
                return (a.v > b.v) - (a.v < b.v)
            if isinstance(a, str):
                return N["java/lang/String.compareTo:(Ljava/lang/String;)I"](j, a, b)
            return j.call_virtual(a, "compareTo", "(Ljava/lang/Object;)I", b)
        return j.call_fn(j, c, a, b) if not (isinstance(c, JObject) and c.cls not in ("$Comparator", "$Fn") and not isinstance(c, JLambda)) \
            else j.call_virtual(c, "compare", "(Ljava/lang/Object;Ljava/lang/Object;)I", a, b)

    def find(j, o, v):
        """-> (index, found) by binary search"""
        items = o.native["items"]
        lo, hi = 0, len(items)
        while lo < hi:
            mid = (lo + hi) // 2
            c = cmp(j, o, items[mid], v)
            c = c.v if isinstance(c, JBox) else c
            if c == 0:
                return mid, True
            if c < 0:
                lo = mid + 1
            else:
                hi = mid
        return lo, False

    def new(j):
        o = JObject("java/util/TreeSet")
        o.native = {"items": [], "cmp": None}
        return o

    def init(j, o, *a):
        if o.native is None:
            o.native = {"items": [], "cmp": None}
        if a and a[0] is not None:
            if isinstance(a[0], JObject) and isinstance(a[0].native, list):
                for v in a[0].native:
                    add(j, o, v)
            else:
                o.native["cmp"] = a[0]

    def add(j, o, v):
        i, found = find(j, o, v)
        if found:
            return 0
        o.native["items"].insert(i, v)
        return 1

    def remove(j, o, v):
        i, found = find(j, o, v)
        if found:
            o.native["items"].pop(i)
            return 1
        return 0

    def first(j, o):
        if not o.native["items"]:
            j.throw("java/util/NoSuchElementException")
        return o.native["items"][0]

    def last(j, o):
        if not o.native["items"]:
            j.throw("java/util/NoSuchElementException")
        return o.native["items"][-1]

    def as_list(j, items):
        lst = JObject("java/util/ArrayList")
        lst.native = list(items)
        return lst

    N["java/util/TreeSet.<new>"] = new
    N["java/util/TreeSet.<init>"] = init
    N["java/util/TreeSet.add"] = add
    N["java/util/TreeSet.remove"] = remove
    N["java/util/TreeSet.contains"] = lambda j, o, v: 1 if find(j, o, v)[1] else 0
    N["java/util/TreeSet.first"] = first
    N["java/util/TreeSet.last"] = last
    N["java/util/TreeSet.size"] = lambda j, o: len(o.native["items"])
    N["java/util/TreeSet.isEmpty"] = lambda j, o: 0 if o.native["items"] else 1
    N["java/util/TreeSet.clear"] = lambda j, o: o.native["items"].clear()
    def ts_iter(j, o):
        it = JObject("$TreeSetIterator")
        it.native = [o, 0, -1]
        return it

    def ts_next(j, it):
        o, i, _ = it.native
        if i >= len(o.native["items"]):
            j.throw("java/util/NoSuchElementException")
        it.native[1], it.native[2] = i + 1, i
        return o.native["items"][i]

    def ts_remove(j, it):
        o, i, last = it.native
        if last < 0:
            j.throw("java/lang/IllegalStateException")
        o.native["items"].pop(last)
        it.native[1], it.native[2] = last, -1

    N["java/util/TreeSet.iterator"] = ts_iter
    N["$TreeSetIterator.hasNext"] = lambda j, it: 1 if it.native[1] < len(it.native[0].native["items"]) else 0
    N["$TreeSetIterator.next"] = ts_next
    N["$TreeSetIterator.remove"] = ts_remove
    __import__("jvm_exec").JDK_IFACES["$TreeSetIterator"] = ["java/util/Iterator"]
    N["java/util/TreeSet.stream"] = lambda j, o: N["java/util/ArrayList.stream"](j, as_list(j, o.native["items"]))
    N["java/util/TreeSet.comparator"] = lambda j, o: o.native["cmp"]
    for k in list(N):
        if k.startswith("java/util/TreeSet.") and not k.endswith(("<new>", "<init>")):
            N.setdefault("java/util/SortedSet" + k[len("java/util/TreeSet"):], N[k])
    sup = __import__("jvm_exec").JDK_SUPER
    ifs = __import__("jvm_exec").JDK_IFACES
    sup["java/util/TreeSet"] = "java/util/AbstractSet"
    ifs["java/util/TreeSet"] = ["java/util/NavigableSet", "java/util/SortedSet", "java/util/Set", "java/util/Collection"]


def install_env(jvm):
    N = jvm.natives

    def inert(cls):
        def f(j, *a):
            o = JObject(cls)
            o.native = {}
            return o

        return f

    N["java/lang/Runtime.getRuntime"] = inert("java/lang/Runtime")
    N["java/lang/Runtime.availableProcessors"] = lambda j, o: 8
    N["java/lang/Runtime.maxMemory"] = lambda j, o: 1 << 34
    N["java/lang/Runtime.totalMemory"] = lambda j, o: 1 << 33
    N["java/lang/Runtime.freeMemory"] = lambda j, o: 1 << 32
    N["java/lang/System.currentTimeMillis"] = lambda j: 0
    N["java/lang/System.nanoTime"] = lambda j: 0
    N["java/lang/System.getProperty"] = lambda j, k, *d: {"user.home": "/tmp", "user.dir": "/tmp", "line.separator": "\n", "file.separator": "/"}.get(k, d[0] if d else None)
    N["java/lang/System.lineSeparator"] = lambda j: "\n"
    N["java/lang/Thread.currentThread"] = inert("java/lang/Thread")
    N["java/lang/Thread.getName"] = lambda j, o: "main"
    N["java/lang/Thread.getId"] = lambda j, o: 1

    for c, wrap in (("java/util/concurrent/atomic/AtomicInteger", i32), ("java/util/concurrent/atomic/AtomicLong", i64)):
        def mk(cls):
            def new(j):
                o = JObject(cls)
                o.native = [0]
                return o

            return new

        N[f"{c}.<new>"] = mk(c)
        N[f"{c}.<init>"] = lambda j, o, *a: o.native.__setitem__(0, a[0] if a else 0)
        N[f"{c}.get"] = lambda j, o: o.native[0]
        N[f"{c}.set"] = lambda j, o, v: o.native.__setitem__(0, v)
        N[f"{c}.intValue"] = lambda j, o: i32(o.native[0])
        N[f"{c}.longValue"] = lambda j, o: i64(o.native[0])
        N[f"{c}.incrementAndGet"] = (lambda w: lambda j, o: (o.native.__setitem__(0, w(o.native[0] + 1)), o.native[0])[1])(wrap)
        N[f"{c}.getAndIncrement"] = (lambda w: lambda j, o: (o.native[0], o.native.__setitem__(0, w(o.native[0] + 1)))[0])(wrap)
        N[f"{c}.addAndGet"] = (lambda w: lambda j, o, d: (o.native.__setitem__(0, w(o.native[0] + d)), o.native[0])[1])(wrap)
        N[f"{c}.getAndAdd"] = (lambda w: lambda j, o, d: (o.native[0], o.native.__setitem__(0, w(o.native[0] + d)))[0])(wrap)
        N[f"{c}.getAndSet"] = lambda j, o, v: (o.native[0], o.native.__setitem__(0, v))[0]
        N[f"{c}.decrementAndGet"] = (lambda w: lambda j, o: (o.native.__setitem__(0, w(o.native[0] - 1)), o.native[0])[1])(wrap)

    # java.text.DecimalFormat: only patterns made of '#', '0', ',', '.' (+ a ' %' suffix); RoundingMode.HALF_EVEN on the exact binary value
    def df_new(j):
        o = JObject("java/text/DecimalFormat")
        o.native = {"pattern": "#"}
        return o

    def df_init(j, o, pattern="#", *a):
        o.native["pattern"] = pattern  # checked when something is formatted with it

    def df_format(j, o, v):
        import decimal

        pat = o.native["pattern"]
        # a suffix of literal blanks and one '%' (the value is multiplied by 100 in double arithmetic first, as DecimalFormat does: `number *= multiplier`)
        suffix = ""
        while pat and pat[-1] in " %":
            suffix = pat[-1] + suffix
            pat = pat[:-1]
        if suffix.count("%") > 1 or any(ch not in "#0,." for ch in pat):
            raise Unsupported(f"DecimalFormat pattern {o.native['pattern']!r}")
        if "%" in suffix:
            v = (v.v if isinstance(v, JBox) else v) * 100.0
        ip, _, fp = pat.partition(".")
        max_frac, min_frac = len(fp), fp.count("0")
        min_int = ip.replace(",", "").count("0")
        grouping = len(ip) - ip.rfind(",") - 1 if "," in ip else 0
        if isinstance(v, JBox):
            v = v.v
        d = decimal.Decimal(v)  # exact value of the double (a float argument was widened exactly)
        q = d.quantize(decimal.Decimal(1).scaleb(-max_frac), rounding=decimal.ROUND_HALF_EVEN)
        neg = q < 0
        q = abs(q)
        s = f"{q:f}"
        i, _, f = s.partition(".")
        f = f.rstrip("0")
        f = f.ljust(min_frac, "0")
        i = i.lstrip("0").rjust(min_int, "0")
        if grouping and len(i) > grouping:
            parts = []
            while len(i) > grouping:
                parts.insert(0, i[-grouping:])
                i = i[:-grouping]
            i = ",".join([i] + parts)
        out = i + ("." + f if f else "")
        if not out:
            out = "0"
        if neg and any(ch in "123456789" for ch in out):
            out = "-" + out
        return out + suffix

    N["java/text/DecimalFormat.<new>"] = df_new
    N["java/text/DecimalFormat.<init>"] = df_init
    N["java/text/DecimalFormat.format:(D)Ljava/lang/String;"] = df_format
    N["java/text/DecimalFormat.format:(J)Ljava/lang/String;"] = df_format
    N["java/text/DecimalFormat.format:(Ljava/lang/Object;)Ljava/lang/String;"] = df_format   # Float / Double / Integer boxes
    N["java/text/NumberFormat.format:(D)Ljava/lang/String;"] = df_format
    N["java/text/NumberFormat.format:(J)Ljava/lang/String;"] = df_format
    N["java/text/Format.format:(Ljava/lang/Object;)Ljava/lang/String;"] = df_format


def install_enumset(jvm):
    """java.util.EnumSet: iteration order is the natural order of the constants (specified), kept as an ordinal-sorted list"""
    N = jvm.natives

    def mk(items=()):
        o = JObject("java/util/EnumSet")
        o.native = sorted(set(items), key=lambda e: e.f["$ordinal"])
        return o

    def add(j, o, e):
        if any(x is e for x in o.native):
            return 0
        o.native.append(e)
        o.native.sort(key=lambda x: x.f["$ordinal"])
        return 1

    def remove(j, o, e):
        for k, x in enumerate(o.native):
            if x is e:
                del o.native[k]
                return 1
        return 0

    def all_of(j, cls):
        vals = j.call_static(cls.native, "values", f"()[L{cls.native};")
        return mk(vals.a)

    N["java/util/EnumSet.noneOf"] = lambda j, cls: mk()
    N["java/util/EnumSet.allOf"] = all_of
    N["java/util/EnumSet.of"] = lambda j, *a: mk([x for x in a if not isinstance(x, JArray)] + [y for x in a if isinstance(x, JArray) for y in x.a])
    N["java/util/EnumSet.copyOf"] = lambda j, c: mk(c.native)
    N["java/util/EnumSet.clone"] = lambda j, o: mk(o.native)
    N["java/util/EnumSet.add"] = add
    N["java/util/EnumSet.remove"] = remove
    N["java/util/EnumSet.contains"] = lambda j, o, e: 1 if any(x is e for x in o.native) else 0
    N["java/util/EnumSet.size"] = lambda j, o: len(o.native)
    N["java/util/EnumSet.isEmpty"] = lambda j, o: 0 if o.native else 1
    N["java/util/EnumSet.clear"] = lambda j, o: o.native.clear()
    N["java/util/EnumSet.addAll"] = lambda j, o, c: 1 if [add(j, o, e) for e in list(c.native)].count(1) else 0
    N["java/util/EnumSet.iterator"] = N["java/util/ArrayList.iterator"]
    N["java/util/EnumSet.stream"] = N["java/util/ArrayList.stream"]
    N["java/util/EnumSet.forEach"] = N["java/util/ArrayList.forEach"]
    ex = __import__("jvm_exec")
    ex.JDK_SUPER["java/util/EnumSet"] = "java/util/AbstractSet"
    ex.JDK_IFACES["java/util/EnumSet"] = ["java/util/Set", "java/util/Collection", "java/lang/Iterable"]
