#!/usr/bin/env python3
"""A small JVM bytecode INTERPRETER: executes methods of the reference's own class files.

Test infrastructure for pinning the oracle (build container only; neither this tool nor the reference's jars travel
to the GPU box).  The reference ships the hot path as bytecode only and the image has no JVM, so the only way to obtain
answers computed BY THE REFERENCE'S CODE is to execute that bytecode here.  tools/make_ref_exec.py drives this
interpreter over the reference's leaf methods and commits inputs + outputs as tests/golden/ref_exec_*.json.

What is interpreted: every instruction of every method of every class found in the jars given to `JVM(jars)` --
constant pool, fields, statics and <clinit>, virtual / interface dispatch, exceptions, lambdas (invokedynamic through
LambdaMetafactory) and string concatenation recipes.  Nothing of the reference is restated here.

What is NOT in the jars is the JDK itself (java.*).  The interpreter supplies it in three clearly separated tiers and
records, per top-level call, which natives ran (`JVM.natives_used`), so that every fixture can say what it rests on:
  tier A  language-level only: java/lang/Object.<init>, arrays, arithmetic -- no library behaviour at all;
  tier B  java.lang value classes whose behaviour the Java SE specification fixes exactly: String, StringBuilder,
          Math, boxing (Integer / Long / Float / Double / Byte / Character / Boolean), System.arraycopy, Arrays.fill /
          copyOf / copyOfRange, Objects, enums' own bytecode;
  tier C  ordered java.util containers (ArrayList, ArrayDeque, LinkedList as Python lists) and MEMBERSHIP-ONLY sets
          (contains / add on a Python set: no iteration order is ever observable through them).
Hash-ordered iteration (HashMap / HashSet / fastutil / eclipse-collections iteration) is deliberately absent: a method
that needs it raises `Unsupported`, and the fixture generator lists it as excluded.
"""
import math
import struct
import sys
import zipfile

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from classdis import ClassFile, OPS  # noqa: E402

TOP = object()  # second slot of a long / double


class Unsupported(Exception):
    pass


class JavaThrow(Exception):
    def __init__(self, obj):
        Exception.__init__(self, getattr(obj, "cls", str(obj)))
        self.obj = obj
        self.trace = []  # innermost first: the Java methods the exception passed through


class JObject:
    __slots__ = ("cls", "f", "native")

    def __init__(self, cls):
        self.cls = cls  # internal name
        self.f = {}
        self.native = None

    def __repr__(self):
        return f"<{self.cls} {self.f if self.native is None else self.native}>"


class JArray:
    __slots__ = ("etype", "a")

    def __init__(self, etype, a):
        self.etype = etype  # element descriptor: 'I', 'J', 'B', '[J', 'Ljava/lang/String;' ...
        self.a = a

    @property
    def cls(self):
        return "[" + self.etype

    def __repr__(self):
        return f"<{self.etype}[{len(self.a)}]>"


class JBox:
    """java.lang.Integer & co: immutable value boxes"""
    __slots__ = ("cls", "v")

    def __init__(self, cls, v):
        self.cls, self.v = cls, v

    def __repr__(self):
        return f"<{self.cls.split('/')[-1]} {self.v}>"

    def __eq__(self, o):
        return isinstance(o, JBox) and o.cls == self.cls and o.v == self.v

    def __hash__(self):
        return hash((self.cls, self.v))


class JLambda:
    __slots__ = ("cls", "iface", "sam", "kind", "owner", "name", "desc", "captured", "inst_desc")

    def __init__(self, iface, sam, kind, owner, name, desc, captured, inst_desc):
        self.cls = "$Lambda"
        self.iface, self.sam, self.kind, self.owner, self.name, self.desc = iface, sam, kind, owner, name, desc
        self.captured, self.inst_desc = captured, inst_desc


def i32(v):
    v &= 0xFFFFFFFF
    return v - 0x100000000 if v & 0x80000000 else v


def i64(v):
    v &= 0xFFFFFFFFFFFFFFFF
    return v - 0x10000000000000000 if v & 0x8000000000000000 else v


def f32(v):
    try:
        return struct.unpack("f", struct.pack("f", v))[0]
    except OverflowError:
        return math.inf if v > 0 else -math.inf


def parse_desc(desc):
    """'(IJ[BLx;)V' -> ([arg descriptors], return descriptor)"""
    assert desc[0] == "("
    i, args = 1, []
    while desc[i] != ")":
        j = i
        while desc[j] == "[":
            j += 1
        if desc[j] == "L":
            j = desc.index(";", j)
        args.append(desc[i:j + 1])
        i = j + 1
    return args, desc[i + 1:]


def default_value(d):
    if d in ("I", "B", "S", "C", "Z", "J"):
        return 0
    if d in ("F", "D"):
        return 0.0
    return None


class Method:
    __slots__ = ("cls", "acc", "name", "desc", "args", "ret", "max_locals", "code", "exc", "lines", "nargs_slots", "ops", "hit")

    def __init__(self, jc, m):
        self.cls = jc
        self.acc, self.name, self.desc, attrs = m
        self.args, self.ret = parse_desc(self.desc)
        self.code = None
        self.ops = None
        self.exc, self.lines = [], {}
        self.hit = None     # bytearray over the code's pcs: 1 = this instruction was executed (coverage, see JVM.coverage)
        for an, data in attrs:
            if an == "Code":
                self.code = data

    @property
    def static(self):
        return bool(self.acc & 0x0008)

    @property
    def native_or_abstract(self):
        return self.code is None


class JClass:
    def __init__(self, jvm, name, cf):
        self.jvm, self.name, self.cf = jvm, name, cf
        self.super = cf.super
        self.interfaces = cf.interfaces
        self.methods = {}
        for m in cf.methods:
            self.methods[(m[1], m[2])] = Method(self, m)
        self.statics = {}
        self.instance_fields = []
        for acc, fname, fdesc, attrs in cf.fields:
            if acc & 0x0008:
                v = default_value(fdesc)
                for an, d in attrs:
                    if an == "ConstantValue":
                        e = cf.cp[struct.unpack(">H", d)[0]]
                        v = cf.utf(e[1]) if e[0] == "String" else e[1]
                        if e[0] == "Float":
                            v = f32(v)
                self.statics[fname] = v
            else:
                self.instance_fields.append((fname, fdesc))
        self.initialized = False
        self.is_enum = bool(cf.access & 0x4000)
        self.is_interface = bool(cf.access & 0x0200)


JDK_SUPER = {
    "java/lang/Object": None, "java/lang/Throwable": "java/lang/Object", "java/lang/Exception": "java/lang/Throwable",
    "java/lang/Error": "java/lang/Throwable", "java/lang/RuntimeException": "java/lang/Exception",
    "java/lang/IllegalArgumentException": "java/lang/RuntimeException", "java/lang/IllegalStateException": "java/lang/RuntimeException",
    "java/lang/IndexOutOfBoundsException": "java/lang/RuntimeException",
    "java/lang/ArrayIndexOutOfBoundsException": "java/lang/IndexOutOfBoundsException",
    "java/lang/StringIndexOutOfBoundsException": "java/lang/IndexOutOfBoundsException",
    "java/lang/NullPointerException": "java/lang/RuntimeException", "java/lang/ArithmeticException": "java/lang/RuntimeException",
    "java/lang/ClassCastException": "java/lang/RuntimeException", "java/lang/NumberFormatException": "java/lang/IllegalArgumentException",
    "java/lang/NegativeArraySizeException": "java/lang/RuntimeException", "java/lang/UnsupportedOperationException": "java/lang/RuntimeException",
    "java/lang/CloneNotSupportedException": "java/lang/Exception", "java/lang/Enum": "java/lang/Object",
    "java/lang/Number": "java/lang/Object", "java/lang/Integer": "java/lang/Number", "java/lang/Long": "java/lang/Number",
    "java/lang/Float": "java/lang/Number", "java/lang/Double": "java/lang/Number", "java/lang/Byte": "java/lang/Number",
    "java/lang/Short": "java/lang/Number", "java/lang/Character": "java/lang/Object", "java/lang/Boolean": "java/lang/Object",
    "java/lang/String": "java/lang/Object", "java/lang/StringBuilder": "java/lang/Object",
    "java/util/ArrayList": "java/util/AbstractList", "java/util/AbstractList": "java/util/AbstractCollection",
    "java/util/AbstractCollection": "java/lang/Object", "java/util/ArrayDeque": "java/util/AbstractCollection",
    "java/util/LinkedList": "java/util/AbstractList", "java/util/NoSuchElementException": "java/lang/RuntimeException",
}
JDK_IFACES = {
    "java/util/ArrayList": ["java/util/List", "java/util/Collection", "java/lang/Iterable", "java/util/RandomAccess"],
    "java/util/LinkedList": ["java/util/List", "java/util/Collection", "java/lang/Iterable", "java/util/Deque", "java/util/Queue"],
    "java/util/ArrayDeque": ["java/util/Deque", "java/util/Queue", "java/util/Collection", "java/lang/Iterable"],
    "java/lang/String": ["java/lang/CharSequence", "java/lang/Comparable"],
    "java/lang/StringBuilder": ["java/lang/CharSequence", "java/lang/Appendable"],
    "java/lang/Integer": ["java/lang/Comparable"], "java/lang/Long": ["java/lang/Comparable"], "java/lang/Float": ["java/lang/Comparable"],
    "java/lang/Double": ["java/lang/Comparable"],
}


class Frame:
    __slots__ = ("m", "locals", "stack", "pc")


class JVM:
    def __init__(self, jars, max_steps=2_000_000_000):
        self.zips = [zipfile.ZipFile(j) for j in jars]
        self.index = {}
        for z in self.zips:
            for n in z.namelist():
                if n.endswith(".class"):
                    self.index.setdefault(n[:-6], z)
        self.classes = {}
        self.natives = {}
        self.natives_used = set()
        self.steps = 0
        self.max_steps = max_steps
        self.depth = 0
        self.intern = {}
        self.hash_order = None  # see jvm_natives.HashStore.cells_for_iteration
        self.hooks = {}  # "cls.name:desc" -> python callable(jvm, args) replacing a method (used for MISSING libraries only)
        from jvm_natives import install, install_enumset, install_env, install_hash, install_streams, install_treeset  # noqa: E402

        install(self)
        install_streams(self)
        install_hash(self)
        install_env(self)
        install_enumset(self)
        install_treeset(self)

    # ---- classes ---------------------------------------------------------------------------------------------------
    def has_class(self, name):
        return name in self.index

    def load(self, name):
        jc = self.classes.get(name)
        if jc is None:
            z = self.index.get(name)
            if z is None:
                raise Unsupported(f"class {name} is not in the jars (JDK or missing dependency)")
            jc = JClass(self, name, ClassFile(z.read(name + ".class")))
            self.classes[name] = jc
        return jc

    def init_class(self, name):
        jc = self.load(name)
        if not jc.initialized:
            jc.initialized = True
            if jc.super and self.has_class(jc.super):
                self.init_class(jc.super)
            m = jc.methods.get(("<clinit>", "()V"))
            if m is not None and f"{name}.<clinit>:()V" not in self.hooks:
                self.run(m, [])
        return jc

    def superclass(self, name):
        if name in self.index:
            return self.load(name).super
        if name.startswith("["):
            return "java/lang/Object"
        if name in JDK_SUPER:
            return JDK_SUPER[name]
        return "java/lang/Object"

    def interfaces_of(self, name):
        if name in self.index:
            return self.load(name).interfaces
        return JDK_IFACES.get(name, [])

    def is_subclass(self, name, target):
        if target == "java/lang/Object" or name == target:
            return True
        seen = set()
        stack = [name]
        while stack:
            n = stack.pop()
            if n is None or n in seen:
                continue
            seen.add(n)
            if n == target:
                return True
            stack.append(self.superclass(n))
            stack.extend(self.interfaces_of(n))
        return False

    def class_of(self, v):
        if isinstance(v, str):
            return "java/lang/String"
        if isinstance(v, JLambda):
            return v.iface
        return v.cls

    def instance_of(self, v, target):
        if v is None:
            return False
        if isinstance(v, JLambda):
            return target in (v.iface, "java/lang/Object")
        c = self.class_of(v)
        if c.startswith("[") and target.startswith("["):
            return c == target or target == "[Ljava/lang/Object;"
        return self.is_subclass(c, target)

    def new_object(self, cname):
        o = JObject(cname)
        n = cname
        while n is not None and n in self.index:
            jc = self.load(n)
            for fname, fdesc in jc.instance_fields:
                o.f.setdefault(fname, default_value(fdesc))
            n = jc.super
        return o

    def find_method(self, cname, name, desc):
        """resolution along the superclass chain, then default methods of interfaces; -> Method or native key or None"""
        n = cname
        seen_ifaces = []
        if cname.startswith("[") and name == "clone":
            return "[.clone"
        while n is not None:
            if n in self.index:
                jc = self.load(n)
                m = jc.methods.get((name, desc))
                if m is not None and not (m.code is None and (m.acc & 0x0400)):
                    return m
                seen_ifaces.extend(jc.interfaces)
                n = jc.super
            else:
                key = f"{n}.{name}:{desc}"
                if key in self.natives:
                    return key
                key2 = f"{n}.{name}"
                if key2 in self.natives:
                    return key2
                if f"{n}.*" in self.natives:
                    return f"{n}.*"
                seen_ifaces.extend(JDK_IFACES.get(n, []))
                n = JDK_SUPER.get(n)
        # default methods
        done = set()
        while seen_ifaces:
            i = seen_ifaces.pop()
            if i in done:
                continue
            done.add(i)
            if i in self.index:
                jc = self.load(i)
                m = jc.methods.get((name, desc))
                if m is not None and m.code is not None:
                    return m
                seen_ifaces.extend(jc.interfaces)
            else:
                for key in (f"{i}.{name}:{desc}", f"{i}.{name}", f"{i}.*"):
                    if key in self.natives:
                        return key
        return None

    def throw(self, cname, msg=None):
        o = JObject(cname)
        o.f["message"] = msg
        raise JavaThrow(o)

    # ---- calls -----------------------------------------------------------------------------------------------------
    def call_native(self, key, args):
        self.natives_used.add(key)
        return self.natives[key](self, *args)

    def invoke(self, target, args):
        """target: Method or native key"""
        if isinstance(target, str):
            return self.call_native(target, args)
        hook = self.hooks.get(f"{target.cls.name}.{target.name}:{target.desc}")
        if hook is None:
            hook = self.hooks.get(f"{target.cls.name}.*")
        if hook is not None:
            self.natives_used.add(f"hook:{target.cls.name}.{target.name}")
            return hook(self, *args)
        return self.run(target, args)

    def call_static(self, cname, name, desc, *args):
        jc = self.init_class(cname)
        m = jc.methods[(name, desc)]
        return self.run(m, list(args))

    def call_virtual(self, obj, name, desc, *args):
        t = self.find_method(self.class_of(obj), name, desc)
        if t is None:
            raise Unsupported(f"no method {self.class_of(obj)}.{name}:{desc}")
        return self.invoke(t, [obj] + list(args))

    def new(self, cname, desc="()V", *args):
        self.init_class(cname)
        o = self.new_object(cname)
        self.run(self.load(cname).methods[("<init>", desc)], [o] + list(args))
        return o

    def call_lambda(self, lam, args):
        """invoke the functional-interface method of a JLambda"""
        kind = lam.kind
        allargs = list(lam.captured) + list(args)
        pargs, pret = parse_desc(lam.desc)
        if kind == 6:  # invokestatic
            self.init_class(lam.owner) if lam.owner in self.index else None
            t = self.find_method(lam.owner, lam.name, lam.desc)
            call = self._adapt(allargs, pargs)
        elif kind in (5, 7, 9):  # invokevirtual / invokespecial / invokeinterface: receiver first
            recv = allargs[0]
            if recv is None:
                self.throw("java/lang/NullPointerException")
            rc = lam.owner if kind == 7 else self.class_of(recv)
            if isinstance(recv, JLambda):
                return self._box_ret(self.call_lambda(recv, self._adapt(allargs[1:], pargs)), pret, lam)
            t = self.find_method(rc, lam.name, lam.desc)
            call = [recv] + self._adapt(allargs[1:], pargs)
        elif kind == 8 and lam.owner not in self.index:  # constructor reference to a JDK class held as a native
            o = self.call_native(f"{lam.owner}.<new>", [])
            key = f"{lam.owner}.<init>:{lam.desc}"
            self.call_native(key if key in self.natives else f"{lam.owner}.<init>", [o] + self._adapt(allargs, pargs))
            return o
        elif kind == 8:  # newinvokespecial
            o = self.new_object(lam.owner)
            self.init_class(lam.owner)
            self.run(self.load(lam.owner).methods[("<init>", lam.desc)], [o] + self._adapt(allargs, pargs))
            return o
        else:
            raise Unsupported(f"method handle kind {kind}")
        if t is None:
            raise Unsupported(f"lambda target {lam.owner}.{lam.name}:{lam.desc}")
        return self._box_ret(self.invoke(t, call), pret, lam)

    _UNBOX = {"I": "java/lang/Integer", "J": "java/lang/Long", "F": "java/lang/Float", "D": "java/lang/Double", "B": "java/lang/Byte",
              "S": "java/lang/Short", "C": "java/lang/Character", "Z": "java/lang/Boolean"}

    def _adapt(self, vals, descs):
        out = []
        for v, d in zip(vals, descs):
            if d in self._UNBOX and isinstance(v, JBox):
                v = v.v
            elif d not in self._UNBOX and not isinstance(v, (JBox, JObject, JArray, JLambda, str, type(None))):
                raise Unsupported("boxing of a primitive captured value")
            out.append(v)
        return out

    def _box_ret(self, v, pret, lam):
        iret = parse_desc(lam.inst_desc)[1]
        if pret in self._UNBOX and iret not in self._UNBOX and iret != "V":
            return JBox(self._UNBOX[pret], v)
        if pret not in self._UNBOX and iret in self._UNBOX and isinstance(v, JBox):
            return v.v
        return v

    # ---- decoding --------------------------------------------------------------------------------------------------
    def _decode(self, m):
        cf = m.cls.cf
        data = m.code
        max_stack, max_locals, clen = struct.unpack_from(">HHI", data, 0)
        code = data[8:8 + clen]
        p = 8 + clen
        n_exc = struct.unpack_from(">H", data, p)[0]
        p += 2
        exc = []
        for _ in range(n_exc):
            s, e, h, ct = struct.unpack_from(">HHHH", data, p)
            p += 8
            exc.append((s, e, h, cf.cname(ct) if ct else None))
        m.max_locals = max_locals
        m.exc = exc
        # LineNumberTable of the Code attribute: pc -> source line (coverage by source line, JVM.coverage)
        lnt = []
        n_attr = struct.unpack_from(">H", data, p)[0]
        p += 2
        for _ in range(n_attr):
            ni, alen = struct.unpack_from(">HI", data, p)
            p += 6
            if cf.utf(ni) == "LineNumberTable":
                cnt = struct.unpack_from(">H", data, p)[0]
                lnt += [struct.unpack_from(">HH", data, p + 2 + 4 * k) for k in range(cnt)]
            p += alen
        lnt.sort()
        m.lines = lnt
        m.hit = bytearray(clen)
        ops = {}
        pc, n = 0, len(code)
        while pc < n:
            op = code[pc]
            nm, kind = OPS[op]
            start = pc
            pc += 1
            arg = None
            if kind == "s1":
                arg = struct.unpack_from(">b", code, pc)[0]
                pc += 1
            elif kind == "s2":
                arg = struct.unpack_from(">h", code, pc)[0]
                pc += 2
            elif kind == "u1":
                arg = code[pc]
                pc += 1
            elif kind == "cp1":
                arg = code[pc]
                pc += 1
            elif kind == "cp2":
                arg = struct.unpack_from(">H", code, pc)[0]
                pc += 2
            elif kind == "lv1":
                arg = code[pc]
                pc += 1
            elif kind.startswith("lvimp"):
                arg = int(kind[-1])
                nm = nm[:-2]
            elif kind == "iinc":
                arg = (code[pc], struct.unpack_from(">b", code, pc + 1)[0])
                pc += 2
            elif kind == "br2":
                arg = start + struct.unpack_from(">h", code, pc)[0]
                pc += 2
            elif kind == "br4":
                arg = start + struct.unpack_from(">i", code, pc)[0]
                pc += 4
            elif kind in ("ii", "id"):
                arg = struct.unpack_from(">H", code, pc)[0]
                pc += 4
            elif kind == "mana":
                arg = (struct.unpack_from(">H", code, pc)[0], code[pc + 2])
                pc += 3
            elif kind == "wide":
                op2 = code[pc]
                nm = OPS[op2][0]
                idx = struct.unpack_from(">H", code, pc + 1)[0]
                if op2 == 132:
                    arg = (idx, struct.unpack_from(">h", code, pc + 3)[0])
                    pc += 5
                else:
                    arg = idx
                    pc += 3
            elif kind == "tsw":
                pc = (pc + 3) & ~3
                dflt, lo, hi = struct.unpack_from(">iii", code, pc)
                pc += 12
                tg = {}
                for k in range(hi - lo + 1):
                    tg[lo + k] = start + struct.unpack_from(">i", code, pc)[0]
                    pc += 4
                arg = (start + dflt, tg)
                nm = "switch"
            elif kind == "lsw":
                pc = (pc + 3) & ~3
                dflt, np_ = struct.unpack_from(">ii", code, pc)
                pc += 8
                tg = {}
                for k in range(np_):
                    mv, off = struct.unpack_from(">ii", code, pc)
                    tg[mv] = start + off
                    pc += 8
                arg = (start + dflt, tg)
                nm = "switch"
            # resolve constant-pool operands once
            if nm in ("ldc", "ldc_w", "ldc2_w"):
                e = cf.cp[arg]
                if e[0] == "String":
                    arg = ("S", cf.utf(e[1]))
                elif e[0] == "Class":
                    arg = ("C", cf.utf(e[1]))
                elif e[0] == "Float":
                    arg = ("F", f32(e[1]))
                elif e[0] in ("Int", "Long", "Double"):
                    arg = (e[0][0], e[1])
                else:
                    raise Unsupported(f"ldc of {e[0]}")
                nm = "ldc2" if e[0] in ("Long", "Double") else "ldc"
            elif nm in ("getstatic", "putstatic", "getfield", "putfield"):
                e = cf.cp[arg]
                nt = cf.cp[e[2]]
                arg = (cf.cname(e[1]), cf.utf(nt[1]), cf.utf(nt[2]))
            elif nm in ("invokevirtual", "invokespecial", "invokestatic", "invokeinterface"):
                e = cf.cp[arg]
                nt = cf.cp[e[2]]
                d = cf.utf(nt[2])
                a, r = parse_desc(d)
                arg = (cf.cname(e[1]), cf.utf(nt[1]), d, a, r)
            elif nm == "invokedynamic":
                e = cf.cp[arg]
                nt = cf.cp[e[2]]
                arg = (e[1], cf.utf(nt[1]), cf.utf(nt[2]))
            elif nm in ("new", "anewarray", "checkcast", "instanceof"):
                arg = cf.cname(arg)
            elif nm == "multianewarray":
                arg = (cf.cname(arg[0]), arg[1])
            ops[start] = (nm, arg, pc)
        m.ops = ops

    # ---- the interpreter loop ------------------------------------------------------------------------------------
    def run(self, m, args):
        if m.code is None:
            raise Unsupported(f"abstract / native method {m.cls.name}.{m.name}:{m.desc}")
        if m.ops is None:
            self._decode(m)
        self.depth += 1
        if self.depth > 400:
            raise Unsupported("call depth > 400")
        try:
            return self._run(m, args)
        except JavaThrow as e:
            if len(e.trace) < 30:
                e.trace.append(f"{m.cls.name}.{m.name}")
            raise
        finally:
            self.depth -= 1

    def _run(self, m, args):
        cf = m.cls.cf
        loc = [None] * (m.max_locals + 2)
        k = 0
        descs = m.args if m.static else ["L"] + m.args
        for v, d in zip(args, descs):
            loc[k] = v
            k += 1
            if d in ("J", "D"):
                loc[k] = TOP
                k += 1
        st = []
        push, pop = st.append, st.pop
        ops = m.ops
        hit = m.hit
        pc = 0
        while True:
            self.steps += 1
            if self.steps > self.max_steps:
                raise Unsupported("step budget exhausted")
            nm, arg, nxt = ops[pc]
            hit[pc] = 1
            try:
                # ---- loads / stores / constants
                if nm in ("iload", "aload", "fload"):
                    push(loc[arg])
                elif nm in ("lload", "dload"):
                    push(loc[arg]); push(TOP)
                elif nm in ("istore", "astore", "fstore"):
                    loc[arg] = pop()
                elif nm in ("lstore", "dstore"):
                    pop(); loc[arg] = pop(); loc[arg + 1] = TOP
                elif nm == "iinc":
                    loc[arg[0]] = i32(loc[arg[0]] + arg[1])
                elif nm.startswith("iconst_"):
                    push(-1 if nm.endswith("m1") else int(nm[-1]))
                elif nm in ("bipush", "sipush"):
                    push(arg)
                elif nm == "aconst_null":
                    push(None)
                elif nm.startswith("lconst_"):
                    push(int(nm[-1])); push(TOP)
                elif nm.startswith("fconst_"):
                    push(float(nm[-1]))
                elif nm.startswith("dconst_"):
                    push(float(nm[-1])); push(TOP)
                elif nm == "ldc":
                    t, v = arg
                    if t == "C":
                        v = self.class_object(v)
                    push(v)
                elif nm == "ldc2":
                    push(arg[1]); push(TOP)
                # ---- arrays
                elif nm in ("iaload", "baload", "caload", "saload", "aaload", "faload"):
                    i = pop(); a = pop()
                    if a is None:
                        self.throw("java/lang/NullPointerException")
                    if not 0 <= i < len(a.a):
                        self.throw("java/lang/ArrayIndexOutOfBoundsException", f"Index {i} out of bounds for length {len(a.a)}")
                    push(a.a[i])
                elif nm in ("laload", "daload"):
                    i = pop(); a = pop()
                    if a is None:
                        self.throw("java/lang/NullPointerException")
                    if not 0 <= i < len(a.a):
                        self.throw("java/lang/ArrayIndexOutOfBoundsException", f"Index {i} out of bounds for length {len(a.a)}")
                    push(a.a[i]); push(TOP)
                elif nm in ("iastore", "aastore", "fastore", "bastore", "castore", "sastore"):
                    v = pop(); i = pop(); a = pop()
                    if a is None:
                        self.throw("java/lang/NullPointerException")
                    if not 0 <= i < len(a.a):
                        self.throw("java/lang/ArrayIndexOutOfBoundsException", f"Index {i} out of bounds for length {len(a.a)}")
                    if nm == "bastore":
                        v = (v & 1) if a.etype == "Z" else ((v + 128) & 255) - 128
                    elif nm == "castore":
                        v &= 0xFFFF
                    elif nm == "sastore":
                        v = ((v + 32768) & 65535) - 32768
                    a.a[i] = v
                elif nm in ("lastore", "dastore"):
                    pop(); v = pop(); i = pop(); a = pop()
                    if a is None:
                        self.throw("java/lang/NullPointerException")
                    if not 0 <= i < len(a.a):
                        self.throw("java/lang/ArrayIndexOutOfBoundsException", f"Index {i} out of bounds for length {len(a.a)}")
                    a.a[i] = v
                elif nm == "arraylength":
                    a = pop()
                    if a is None:
                        self.throw("java/lang/NullPointerException")
                    push(len(a.a))
                elif nm == "newarray":
                    n = pop()
                    if n < 0:
                        self.throw("java/lang/NegativeArraySizeException")
                    et = {4: "Z", 5: "C", 6: "F", 7: "D", 8: "B", 9: "S", 10: "I", 11: "J"}[arg]
                    push(JArray(et, [default_value(et)] * n))
                elif nm == "anewarray":
                    n = pop()
                    if n < 0:
                        self.throw("java/lang/NegativeArraySizeException")
                    push(JArray(arg if arg.startswith("[") else "L" + arg + ";", [None] * n))
                elif nm == "multianewarray":
                    cname, dims = arg
                    counts = [pop() for _ in range(dims)][::-1]

                    def mk(desc, cs):
                        et = desc[1:]
                        if len(cs) == 1:
                            return JArray(et, [default_value(et)] * cs[0])
                        return JArray(et, [mk(et, cs[1:]) for _ in range(cs[0])])

                    push(mk(cname, counts))
                # ---- stack
                elif nm == "pop":
                    pop()
                elif nm == "pop2":
                    pop(); pop()
                elif nm == "dup":
                    push(st[-1])
                elif nm == "dup_x1":
                    a = pop(); b = pop(); push(a); push(b); push(a)
                elif nm == "dup_x2":
                    a = pop(); b = pop(); c = pop(); push(a); push(c); push(b); push(a)
                elif nm == "dup2":
                    a = st[-1]; b = st[-2]; push(b); push(a)
                elif nm == "dup2_x1":
                    a = pop(); b = pop(); c = pop(); push(b); push(a); push(c); push(b); push(a)
                elif nm == "dup2_x2":
                    a = pop(); b = pop(); c = pop(); d = pop(); push(b); push(a); push(d); push(c); push(b); push(a)
                elif nm == "swap":
                    a = pop(); b = pop(); push(a); push(b)
                # ---- int arithmetic
                elif nm == "iadd":
                    b = pop(); push(i32(pop() + b))
                elif nm == "isub":
                    b = pop(); push(i32(pop() - b))
                elif nm == "imul":
                    b = pop(); push(i32(pop() * b))
                elif nm == "idiv":
                    b = pop(); a = pop()
                    if b == 0:
                        self.throw("java/lang/ArithmeticException", "/ by zero")
                    q = abs(a) // abs(b)
                    push(i32(q if (a < 0) == (b < 0) else -q))
                elif nm == "irem":
                    b = pop(); a = pop()
                    if b == 0:
                        self.throw("java/lang/ArithmeticException", "/ by zero")
                    r = abs(a) % abs(b)
                    push(i32(-r if a < 0 else r))
                elif nm == "ineg":
                    push(i32(-pop()))
                elif nm == "ishl":
                    b = pop(); push(i32(pop() << (b & 31)))
                elif nm == "ishr":
                    b = pop(); push(pop() >> (b & 31))
                elif nm == "iushr":
                    b = pop(); push(i32((pop() & 0xFFFFFFFF) >> (b & 31)))
                elif nm == "iand":
                    b = pop(); push(pop() & b)
                elif nm == "ior":
                    b = pop(); push(pop() | b)
                elif nm == "ixor":
                    b = pop(); push(pop() ^ b)
                # ---- long arithmetic
                elif nm in ("ladd", "lsub", "lmul", "land", "lor", "lxor"):
                    pop(); b = pop(); pop(); a = pop()
                    if nm == "ladd":
                        r = a + b
                    elif nm == "lsub":
                        r = a - b
                    elif nm == "lmul":
                        r = a * b
                    elif nm == "land":
                        r = a & b
                    elif nm == "lor":
                        r = a | b
                    else:
                        r = a ^ b
                    push(i64(r)); push(TOP)
                elif nm in ("ldiv", "lrem"):
                    pop(); b = pop(); pop(); a = pop()
                    if b == 0:
                        self.throw("java/lang/ArithmeticException", "/ by zero")
                    if nm == "ldiv":
                        q = abs(a) // abs(b)
                        r = q if (a < 0) == (b < 0) else -q
                    else:
                        r = abs(a) % abs(b)
                        r = -r if a < 0 else r
                    push(i64(r)); push(TOP)
                elif nm == "lneg":
                    pop(); push(i64(-pop())); push(TOP)
                elif nm == "lshl":
                    b = pop(); pop(); a = pop(); push(i64(a << (b & 63))); push(TOP)
                elif nm == "lshr":
                    b = pop(); pop(); a = pop(); push(a >> (b & 63)); push(TOP)
                elif nm == "lushr":
                    b = pop(); pop(); a = pop(); push(i64((a & 0xFFFFFFFFFFFFFFFF) >> (b & 63))); push(TOP)
                elif nm == "lcmp":
                    pop(); b = pop(); pop(); a = pop(); push((a > b) - (a < b))
                # ---- float / double arithmetic
                elif nm in ("fadd", "fsub", "fmul", "fdiv", "frem"):
                    b = pop(); a = pop(); push(f32(self._farith(nm[1:], a, b)))
                elif nm == "fneg":
                    push(-pop())
                elif nm in ("dadd", "dsub", "dmul", "ddiv", "drem"):
                    pop(); b = pop(); pop(); a = pop(); push(self._farith(nm[1:], a, b)); push(TOP)
                elif nm == "dneg":
                    pop(); push(-pop()); push(TOP)
                elif nm in ("fcmpl", "fcmpg"):
                    b = pop(); a = pop()
                    push((1 if nm == "fcmpg" else -1) if (a != a or b != b) else (a > b) - (a < b))
                elif nm in ("dcmpl", "dcmpg"):
                    pop(); b = pop(); pop(); a = pop()
                    push((1 if nm == "dcmpg" else -1) if (a != a or b != b) else (a > b) - (a < b))
                # ---- conversions
                elif nm == "i2l":
                    push(TOP)
                elif nm == "i2f":
                    push(f32(float(pop())))
                elif nm == "i2d":
                    push(float(pop())); push(TOP)
                elif nm == "l2i":
                    pop(); push(i32(pop()))
                elif nm == "l2f":
                    pop(); push(f32(float(pop())))
                elif nm == "l2d":
                    pop(); push(float(pop())); push(TOP)
                elif nm == "f2i":
                    push(self._f2int(pop(), 32))
                elif nm == "f2l":
                    push(self._f2int(pop(), 64)); push(TOP)
                elif nm == "f2d":
                    push(TOP)
                elif nm == "d2i":
                    pop(); push(self._f2int(pop(), 32))
                elif nm == "d2l":
                    pop(); push(self._f2int(pop(), 64)); push(TOP)
                elif nm == "d2f":
                    pop(); push(f32(pop()))
                elif nm == "i2b":
                    push(((pop() + 128) & 255) - 128)
                elif nm == "i2c":
                    push(pop() & 0xFFFF)
                elif nm == "i2s":
                    push(((pop() + 32768) & 65535) - 32768)
                # ---- branches
                elif nm == "goto" or nm == "goto_w":
                    pc = arg
                    continue
                elif nm in ("ifeq", "ifne", "iflt", "ifge", "ifgt", "ifle"):
                    v = pop()
                    if (nm == "ifeq" and v == 0) or (nm == "ifne" and v != 0) or (nm == "iflt" and v < 0) or (nm == "ifge" and v >= 0) or \
                            (nm == "ifgt" and v > 0) or (nm == "ifle" and v <= 0):
                        pc = arg
                        continue
                elif nm.startswith("if_icmp"):
                    b = pop(); a = pop()
                    c = nm[7:]
                    if (c == "eq" and a == b) or (c == "ne" and a != b) or (c == "lt" and a < b) or (c == "ge" and a >= b) or \
                            (c == "gt" and a > b) or (c == "le" and a <= b):
                        pc = arg
                        continue
                elif nm in ("if_acmpeq", "if_acmpne"):
                    b = pop(); a = pop()
                    same = a is b or (isinstance(a, str) and isinstance(b, str) and a == b and self.intern.get(a) is not None)
                    if same == (nm == "if_acmpeq"):
                        pc = arg
                        continue
                elif nm == "ifnull":
                    if pop() is None:
                        pc = arg
                        continue
                elif nm == "ifnonnull":
                    if pop() is not None:
                        pc = arg
                        continue
                elif nm == "switch":
                    v = pop()
                    pc = arg[1].get(v, arg[0])
                    continue
                # ---- returns
                elif nm in ("ireturn", "areturn", "freturn"):
                    v = pop()
                    if nm == "ireturn":
                        r = m.ret
                        if r == "B":
                            v = ((v + 128) & 255) - 128
                        elif r == "Z":
                            v &= 1
                        elif r == "C":
                            v &= 0xFFFF
                        elif r == "S":
                            v = ((v + 32768) & 65535) - 32768
                    return v
                elif nm in ("lreturn", "dreturn"):
                    pop()
                    return pop()
                elif nm == "return":
                    return None
                # ---- fields
                elif nm == "getstatic":
                    cname, fname, fdesc = arg
                    v = self.get_static(cname, fname)
                    push(v)
                    if fdesc in ("J", "D"):
                        push(TOP)
                elif nm == "putstatic":
                    cname, fname, fdesc = arg
                    if fdesc in ("J", "D"):
                        pop()
                    self.put_static(cname, fname, pop())
                elif nm == "getfield":
                    cname, fname, fdesc = arg
                    o = pop()
                    if o is None:
                        self.throw("java/lang/NullPointerException", f"getfield {fname}")
                    push(o.f[fname])
                    if fdesc in ("J", "D"):
                        push(TOP)
                elif nm == "putfield":
                    cname, fname, fdesc = arg
                    if fdesc in ("J", "D"):
                        pop()
                    v = pop(); o = pop()
                    if o is None:
                        self.throw("java/lang/NullPointerException", f"putfield {fname}")
                    if fdesc == "B":
                        v = ((v + 128) & 255) - 128
                    o.f[fname] = v
                # ---- objects
                elif nm == "new":
                    if arg in self.index:
                        self.init_class(arg)
                        push(self.new_object(arg))
                    else:
                        push(self.new_jdk_object(arg))
                elif nm == "checkcast":
                    v = st[-1]
                    if v is not None and not self.instance_of(v, arg):
                        self.throw("java/lang/ClassCastException", f"{self.class_of(v)} -> {arg}")
                elif nm == "instanceof":
                    push(1 if self.instance_of(pop(), arg) else 0)
                elif nm == "athrow":
                    o = pop()
                    if o is None:
                        self.throw("java/lang/NullPointerException")
                    raise JavaThrow(o)
                elif nm in ("monitorenter", "monitorexit"):
                    pop()
                elif nm == "nop":
                    pass
                # ---- invocations
                elif nm in ("invokestatic", "invokevirtual", "invokespecial", "invokeinterface"):
                    cname, mname, mdesc, adescs, rdesc = arg
                    cargs = []
                    for d in reversed(adescs):
                        if d in ("J", "D"):
                            pop()
                        cargs.append(pop())
                    cargs.reverse()
                    if nm == "invokestatic":
                        if cname in self.index:
                            self.init_class(cname)
                        t = self.find_method(cname, mname, mdesc)
                        if t is None:
                            raise Unsupported(f"static {cname}.{mname}:{mdesc} (not in the jars, no native)")
                        r = self.invoke(t, cargs)
                    else:
                        recv = pop()
                        if recv is None:
                            self.throw("java/lang/NullPointerException", f"{cname}.{mname} on null")
                        if mname == "<init>" and cname in self._IMMUTABLE:
                            # `new String(..)`: the value only exists now; swap it in for the placeholder `new` pushed
                            key = f"{cname}.<init>:{mdesc}"
                            if key not in self.natives:
                                key = f"{cname}.<init>"
                            val = self.call_native(key, [recv] + cargs)
                            for q in range(len(st)):
                                if st[q] is recv:
                                    st[q] = val
                            for q in range(len(loc)):
                                if loc[q] is recv:
                                    loc[q] = val
                            r = None
                        elif isinstance(recv, JLambda) and mname == recv.sam:
                            r = self.call_lambda(recv, cargs)
                        elif isinstance(recv, JObject) and recv.cls in ("$Comparator", "$Fn") and mname in ("compare", "apply", "test", "accept", "applyAsInt", "get"):
                            r = recv.native(*cargs)
                        else:
                            if nm == "invokespecial":
                                start = cname
                            else:
                                start = self.class_of(recv)
                            t = self.find_method(start, mname, mdesc)
                            if t is None and isinstance(recv, JLambda):
                                t = self.find_method(recv.iface, mname, mdesc)  # default method of the functional interface
                            if t is None:
                                raise Unsupported(f"{nm} {start}.{mname}:{mdesc} declared in {cname} (not in the jars, no native)")
                            r = self.invoke(t, [recv] + cargs)
                    if rdesc != "V":
                        push(r)
                        if rdesc in ("J", "D"):
                            push(TOP)
                elif nm == "invokedynamic":
                    bidx, dname, ddesc = arg
                    adescs, rdesc = parse_desc(ddesc)
                    cargs = []
                    for d in reversed(adescs):
                        if d in ("J", "D"):
                            pop()
                        cargs.append(pop())
                    cargs.reverse()
                    push(self.indy(cf, bidx, dname, adescs, rdesc, cargs))
                else:
                    raise Unsupported(f"opcode {nm}")
            except JavaThrow as jt:
                h = None
                for s, e, hpc, ct in m.exc:
                    if s <= pc < e and (ct is None or self.instance_of(jt.obj, ct)):
                        h = hpc
                        break
                if h is None:
                    raise
                del st[:]
                push(jt.obj)
                pc = h
                continue
            pc = nxt

    @staticmethod
    def _farith(op, a, b):
        if op == "add":
            return a + b
        if op == "sub":
            return a - b
        if op == "mul":
            return a * b
        if op == "div":
            if b == 0:
                if a != a or a == 0:
                    return math.nan
                return math.copysign(math.inf, a) * math.copysign(1.0, b)
            return a / b
        if b == 0 or a in (math.inf, -math.inf) or a != a or b != b:
            return math.nan
        return math.fmod(a, b)

    @staticmethod
    def _f2int(v, bits):
        if v != v:
            return 0
        lo, hi = -(1 << (bits - 1)), (1 << (bits - 1)) - 1
        if v >= hi:
            return hi
        if v <= lo:
            return lo
        return int(v)

    # ---- statics / misc --------------------------------------------------------------------------------------------
    # ---- coverage ---------------------------------------------------------------------------------------------------
    def coverage(self, class_names):
        """{class: {"method:desc": {"lines": [source lines of the method], "hit": [those with an executed instruction],
        "instr": n instructions, "instr_hit": n executed}}} for the named classes of the jars (a class that was never
        loaded is loaded and decoded here so that its lines count as present).  Lines come from each method's
        LineNumberTable; an instruction belongs to the entry with the greatest start pc <= its pc."""
        import bisect
        rep = {}
        for cname in class_names:
            if not self.has_class(cname):
                continue
            jc = self.load(cname)
            cm = {}
            for (name, desc), m in jc.methods.items():
                if m.code is None:
                    continue
                if m.ops is None:
                    self._decode(m)
                if not m.lines:
                    continue
                starts = [pc for pc, _ in m.lines]
                present, hit = set(), set()
                n_hit = 0
                for pc in m.ops:
                    k = bisect.bisect_right(starts, pc) - 1
                    ln = m.lines[max(k, 0)][1]
                    present.add(ln)
                    if m.hit[pc]:
                        hit.add(ln)
                        n_hit += 1
                cm[f"{name}:{desc}"] = {"lines": sorted(present), "hit": sorted(hit), "instr": len(m.ops), "instr_hit": n_hit}
            rep[cname] = cm
        return rep

    def get_static(self, cname, fname):
        n = cname
        while n is not None:
            if n in self.index:
                jc = self.init_class(n)
                if fname in jc.statics:
                    return jc.statics[fname]
                for i in jc.interfaces:
                    if i in self.index:
                        ji = self.init_class(i)
                        if fname in ji.statics:
                            return ji.statics[fname]
                n = jc.super
            else:
                key = f"{n}.{fname}"
                if key in self.natives:
                    self.natives_used.add(key)
                    return self.natives[key](self)
                raise Unsupported(f"static field {cname}.{fname}")
        raise Unsupported(f"static field {cname}.{fname}")

    def put_static(self, cname, fname, v):
        n = cname
        while n is not None and n in self.index:
            jc = self.init_class(n)
            if fname in jc.statics:
                jc.statics[fname] = v
                return
            n = jc.super
        raise Unsupported(f"putstatic {cname}.{fname}")

    def class_object(self, name):
        o = self.intern.get(("class", name))
        if o is None:
            o = JObject("java/lang/Class")
            o.native = name
            self.intern[("class", name)] = o
        return o

    _IMMUTABLE = {"java/lang/String", "java/lang/Integer", "java/lang/Long", "java/lang/Float", "java/lang/Double", "java/lang/Byte",
                  "java/lang/Short", "java/lang/Character", "java/lang/Boolean"}

    def new_jdk_object(self, cname):
        if cname in self._IMMUTABLE:
            from jvm_natives import Uninit

            return Uninit(cname)
        key = f"{cname}.<new>"
        if key not in self.natives:
            if cname in JDK_SUPER and self.is_subclass(cname, "java/lang/Throwable"):
                o = JObject(cname)
                o.f["message"] = None
                return o
            raise Unsupported(f"new {cname} (JDK class without a native)")
        self.natives_used.add(key)
        return self.natives[key](self)

    def indy(self, cf, bidx, dname, adescs, rdesc, cargs):
        mh, bargs = cf.bootstrap[bidx]
        bsm = cf.const(mh)
        if "LambdaMetafactory.metafactory" in bsm or "LambdaMetafactory.altMetafactory" in bsm:
            impl = cf.cp[bargs[1]]
            kind = impl[1]
            ref = cf.cp[impl[2]]
            nt = cf.cp[ref[2]]
            inst = cf.utf(cf.cp[bargs[2]][1])
            return JLambda(rdesc[1:-1], dname, kind, cf.cname(ref[1]), cf.utf(nt[1]), cf.utf(nt[2]), cargs, inst)
        if "StringConcatFactory.makeConcatWithConstants" in bsm:
            self.natives_used.add("java/lang/invoke/StringConcatFactory.makeConcatWithConstants")
            recipe = cf.utf(cf.cp[bargs[0]][1])
            consts = [cf.cp[b] for b in bargs[1:]]
            out, ai, ci = [], 0, 0
            for ch in recipe:
                if ch == "\x01":
                    out.append(self.to_jstring(cargs[ai], adescs[ai]))
                    ai += 1
                elif ch == "\x02":
                    c = consts[ci]
                    ci += 1
                    out.append(cf.utf(c[1]) if c[0] == "String" else str(c[1]))
                else:
                    out.append(ch)
            return "".join(out)
        raise Unsupported(f"bootstrap method {bsm}")

    def to_jstring(self, v, desc=None):
        """String.valueOf semantics"""
        if desc == "Z":
            return "true" if v else "false"
        if desc == "C":
            return chr(v)
        if desc in ("F", "D"):
            return self.natives["$float_to_string"](v, desc == "F")
        if desc in ("I", "J", "B", "S"):
            return str(v)
        if v is None:
            return "null"
        if isinstance(v, str):
            return v
        if isinstance(v, JBox):
            c = v.cls.split("/")[-1]
            if c == "Character":
                return chr(v.v)
            if c == "Boolean":
                return "true" if v.v else "false"
            if c in ("Float", "Double"):
                return self.natives["$float_to_string"](v.v, c == "Float")
            return str(v.v)
        t = self.find_method(self.class_of(v), "toString", "()Ljava/lang/String;")
        if t is None:
            raise Unsupported(f"toString of {self.class_of(v)}")
        return self.invoke(t, [v])

    # ---- conveniences for drivers ------------------------------------------------------------------------------
    def byte_array(self, vals):
        return JArray("B", [((int(v) + 128) & 255) - 128 for v in vals])

    def char_array(self, s):
        return JArray("C", [ord(c) for c in s])

    def long_array(self, vals):
        return JArray("J", [i64(int(v)) for v in vals])
