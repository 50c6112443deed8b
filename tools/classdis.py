#!/usr/bin/env python3
"""Minimal JVM class-file disassembler (javap -c -l -p look-alike).

Test infrastructure only.  The reference ships its hot path as bytecode
(NanoporeBC_UMI_finder-2.1.jar, package com.rw.*) and this image has no JDK, so
this tool is how the oracle's citations (Class.java:Lnn) were read.  Its OUTPUT
is reference content and must never be committed; write it to a scratch dir.

usage: classdis.py Foo.class [method-substring]
"""
import struct
import sys

OPS = {}


def _o(code, name, kind=""):
    OPS[code] = (name, kind)


for i, n in enumerate(
    "nop aconst_null iconst_m1 iconst_0 iconst_1 iconst_2 iconst_3 iconst_4 iconst_5 lconst_0 lconst_1 "
    "fconst_0 fconst_1 fconst_2 dconst_0 dconst_1".split()
):
    _o(i, n)
_o(16, "bipush", "s1")
_o(17, "sipush", "s2")
_o(18, "ldc", "cp1")
_o(19, "ldc_w", "cp2")
_o(20, "ldc2_w", "cp2")
for i, n in enumerate("iload lload fload dload aload".split()):
    _o(21 + i, n, "lv1")
c = 26
for t in "ilfda":
    for k in range(4):
        _o(c, f"{t}load_{k}", f"lvimp{k}")
        c += 1
for i, n in enumerate("iaload laload faload daload aaload baload caload saload".split()):
    _o(46 + i, n)
for i, n in enumerate("istore lstore fstore dstore astore".split()):
    _o(54 + i, n, "lv1")
c = 59
for t in "ilfda":
    for k in range(4):
        _o(c, f"{t}store_{k}", f"lvimp{k}")
        c += 1
for i, n in enumerate("iastore lastore fastore dastore aastore bastore castore sastore".split()):
    _o(79 + i, n)
for i, n in enumerate("pop pop2 dup dup_x1 dup_x2 dup2 dup2_x1 dup2_x2 swap".split()):
    _o(87 + i, n)
c = 96
for op in "add sub mul div rem neg".split():
    for t in "ilfd":
        _o(c, t + op)
        c += 1
for op in "shl shr ushr and or xor".split():
    for t in "il":
        _o(c, t + op)
        c += 1
_o(132, "iinc", "iinc")
for i, n in enumerate("i2l i2f i2d l2i l2f l2d f2i f2l f2d d2i d2l d2f i2b i2c i2s lcmp fcmpl fcmpg dcmpl dcmpg".split()):
    _o(133 + i, n)
for i, n in enumerate(
    "ifeq ifne iflt ifge ifgt ifle if_icmpeq if_icmpne if_icmplt if_icmpge if_icmpgt if_icmple if_acmpeq if_acmpne goto jsr".split()
):
    _o(153 + i, n, "br2")
_o(169, "ret", "lv1")
_o(170, "tableswitch", "tsw")
_o(171, "lookupswitch", "lsw")
for i, n in enumerate("ireturn lreturn freturn dreturn areturn return".split()):
    _o(172 + i, n)
for i, n in enumerate("getstatic putstatic getfield putfield invokevirtual invokespecial invokestatic".split()):
    _o(178 + i, n, "cp2")
_o(185, "invokeinterface", "ii")
_o(186, "invokedynamic", "id")
_o(187, "new", "cp2")
_o(188, "newarray", "u1")
_o(189, "anewarray", "cp2")
_o(190, "arraylength")
_o(191, "athrow")
_o(192, "checkcast", "cp2")
_o(193, "instanceof", "cp2")
_o(194, "monitorenter")
_o(195, "monitorexit")
_o(196, "wide", "wide")
_o(197, "multianewarray", "mana")
_o(198, "ifnull", "br2")
_o(199, "ifnonnull", "br2")
_o(200, "goto_w", "br4")
_o(201, "jsr_w", "br4")


class Reader:
    def __init__(self, b):
        self.b = b
        self.p = 0

    def u1(self):
        v = self.b[self.p]
        self.p += 1
        return v

    def u2(self):
        v = struct.unpack_from(">H", self.b, self.p)[0]
        self.p += 2
        return v

    def u4(self):
        v = struct.unpack_from(">I", self.b, self.p)[0]
        self.p += 4
        return v

    def raw(self, n):
        v = self.b[self.p : self.p + n]
        self.p += n
        return v


class ClassFile:
    def __init__(self, data):
        r = Reader(data)
        assert r.u4() == 0xCAFEBABE
        r.u2()
        self.major = r.u2()
        n = r.u2()
        cp = [None] * n
        i = 1
        while i < n:
            t = r.u1()
            if t == 1:
                cp[i] = ("Utf8", r.raw(r.u2()).decode("utf-8", "replace"))
            elif t == 3:
                cp[i] = ("Int", struct.unpack(">i", r.raw(4))[0])
            elif t == 4:
                cp[i] = ("Float", struct.unpack(">f", r.raw(4))[0])
            elif t == 5:
                cp[i] = ("Long", struct.unpack(">q", r.raw(8))[0])
                i += 1
            elif t == 6:
                cp[i] = ("Double", struct.unpack(">d", r.raw(8))[0])
                i += 1
            elif t == 7:
                cp[i] = ("Class", r.u2())
            elif t == 8:
                cp[i] = ("String", r.u2())
            elif t in (9, 10, 11):
                cp[i] = ({9: "Field", 10: "Method", 11: "IMethod"}[t], r.u2(), r.u2())
            elif t == 12:
                cp[i] = ("NameType", r.u2(), r.u2())
            elif t == 15:
                cp[i] = ("MHandle", r.u1(), r.u2())
            elif t == 16:
                cp[i] = ("MType", r.u2())
            elif t in (17, 18):
                cp[i] = ("Dynamic" if t == 17 else "InvokeDynamic", r.u2(), r.u2())
            elif t in (19, 20):
                cp[i] = ("Module", r.u2())
            else:
                raise ValueError(f"cp tag {t}")
            i += 1
        self.cp = cp
        self.access = r.u2()
        self.this = self.cname(r.u2())
        sup = r.u2()
        self.super = self.cname(sup) if sup else None
        self.interfaces = [self.cname(r.u2()) for _ in range(r.u2())]
        self.fields = [self._member(r) for _ in range(r.u2())]
        self.methods = [self._member(r) for _ in range(r.u2())]
        self.attrs = self._attrs(r)
        self.bootstrap = []
        for name, data in self.attrs:
            if name == "BootstrapMethods":
                rr = Reader(data)
                for _ in range(rr.u2()):
                    mh = rr.u2()
                    args = [rr.u2() for _ in range(rr.u2())]
                    self.bootstrap.append((mh, args))

    def utf(self, i):
        return self.cp[i][1]

    def cname(self, i):
        return self.utf(self.cp[i][1])

    def _attrs(self, r):
        out = []
        for _ in range(r.u2()):
            name = self.utf(r.u2())
            out.append((name, r.raw(r.u4())))
        return out

    def _member(self, r):
        acc = r.u2()
        name = self.utf(r.u2())
        desc = self.utf(r.u2())
        return (acc, name, desc, self._attrs(r))

    def const(self, i):
        e = self.cp[i]
        t = e[0]
        if t == "Utf8":
            return repr(e[1])
        if t in ("Int", "Long"):
            return f"{t.lower()} {e[1]}" + (f" (0x{e[1] & 0xFFFFFFFFFFFFFFFF:x})" if abs(e[1]) > 255 else "")
        if t in ("Float", "Double"):
            return f"{t.lower()} {e[1]!r}"
        if t == "Class":
            return "class " + self.utf(e[1])
        if t == "String":
            return "String " + repr(self.utf(e[1]))
        if t in ("Field", "Method", "IMethod"):
            nt = self.cp[e[2]]
            return f"{self.cname(e[1])}.{self.utf(nt[1])}:{self.utf(nt[2])}"
        if t == "NameType":
            return f"{self.utf(e[1])}:{self.utf(e[2])}"
        if t == "MHandle":
            return f"MH[{e[1]}] " + self.const(e[2])
        if t == "MType":
            return "MT " + self.utf(e[1])
        if t in ("InvokeDynamic", "Dynamic"):
            nt = self.cp[e[2]]
            bs = self.bootstrap[e[1]] if e[1] < len(self.bootstrap) else None
            s = f"#{e[1]} {self.utf(nt[1])}:{self.utf(nt[2])}"
            if bs:
                s += "  BSM=" + self.const(bs[0]).split(":")[0] + " args=[" + "; ".join(self.const(a) for a in bs[1]) + "]"
            return s
        return str(e)


ATYPE = {4: "boolean", 5: "char", 6: "float", 7: "double", 8: "byte", 9: "short", 10: "int", 11: "long"}


def dis_method(cf, m, out):
    acc, name, desc, attrs = m
    out(f"\n  method {name}{desc}  acc=0x{acc:x}")
    for an, data in attrs:
        if an != "Code":
            continue
        r = Reader(data)
        max_stack, max_locals = r.u2(), r.u2()
        code = r.raw(r.u4())
        exc = [(r.u2(), r.u2(), r.u2(), r.u2()) for _ in range(r.u2())]
        cattrs = cf._attrs(r)
        lines, lvt = {}, []
        for cn, cd in cattrs:
            rr = Reader(cd)
            if cn == "LineNumberTable":
                for _ in range(rr.u2()):
                    pc, ln = rr.u2(), rr.u2()
                    lines.setdefault(pc, ln)
            elif cn == "LocalVariableTable":
                for _ in range(rr.u2()):
                    spc, ln, ni, di, slot = rr.u2(), rr.u2(), rr.u2(), rr.u2(), rr.u2()
                    lvt.append((spc, ln, cf.utf(ni), cf.utf(di), slot))

        def lv(slot, pc, store=False):
            for spc, ln, nm, ds, sl in lvt:
                if sl == slot and (spc <= pc < spc + ln or (store and spc - 4 <= pc < spc + ln)):
                    return f"{slot}<{nm}>"
            return str(slot)

        out(f"    stack={max_stack} locals={max_locals}")
        for spc, ln, nm, ds, sl in sorted(lvt, key=lambda x: (x[4], x[0])):
            out(f"    local slot {sl}: {nm} {ds} pc[{spc},{spc + ln})")
        pc = 0
        n = len(code)
        while pc < n:
            op = code[pc]
            nm, kind = OPS.get(op, (f"op{op}", ""))
            start = pc
            pc += 1
            arg = ""
            if kind == "s1":
                arg = str(struct.unpack_from(">b", code, pc)[0])
                pc += 1
            elif kind == "s2":
                arg = str(struct.unpack_from(">h", code, pc)[0])
                pc += 2
            elif kind == "u1":
                arg = ATYPE.get(code[pc], str(code[pc]))
                pc += 1
            elif kind == "cp1":
                arg = cf.const(code[pc])
                pc += 1
            elif kind == "cp2":
                arg = cf.const(struct.unpack_from(">H", code, pc)[0])
                pc += 2
            elif kind == "lv1":
                arg = lv(code[pc], start, "store" in nm)
                pc += 1
            elif kind.startswith("lvimp"):
                arg = "; " + lv(int(kind[-1]), start, "store" in nm)
            elif kind == "iinc":
                arg = f"{lv(code[pc], start)} += {struct.unpack_from('>b', code, pc + 1)[0]}"
                pc += 2
            elif kind == "br2":
                arg = "-> " + str(start + struct.unpack_from(">h", code, pc)[0])
                pc += 2
            elif kind == "br4":
                arg = "-> " + str(start + struct.unpack_from(">i", code, pc)[0])
                pc += 4
            elif kind == "ii":
                arg = cf.const(struct.unpack_from(">H", code, pc)[0])
                pc += 4
            elif kind == "id":
                arg = cf.const(struct.unpack_from(">H", code, pc)[0])
                pc += 4
            elif kind == "mana":
                arg = cf.const(struct.unpack_from(">H", code, pc)[0]) + f" dims={code[pc + 2]}"
                pc += 3
            elif kind == "wide":
                op2 = code[pc]
                nm2 = OPS[op2][0]
                idx = struct.unpack_from(">H", code, pc + 1)[0]
                if op2 == 132:
                    arg = f"{nm2} {lv(idx, start)} += {struct.unpack_from('>h', code, pc + 3)[0]}"
                    pc += 5
                else:
                    arg = f"{nm2} {lv(idx, start, 'store' in nm2)}"
                    pc += 3
            elif kind == "tsw":
                pc = (pc + 3) & ~3
                dflt, lo, hi = struct.unpack_from(">iii", code, pc)
                pc += 12
                tg = []
                for k in range(hi - lo + 1):
                    tg.append(f"{lo + k}->{start + struct.unpack_from('>i', code, pc)[0]}")
                    pc += 4
                arg = f"default->{start + dflt} " + " ".join(tg)
            elif kind == "lsw":
                pc = (pc + 3) & ~3
                dflt, np_ = struct.unpack_from(">ii", code, pc)
                pc += 8
                tg = []
                for k in range(np_):
                    mv, off = struct.unpack_from(">ii", code, pc)
                    tg.append(f"{mv}->{start + off}")
                    pc += 8
                arg = f"default->{start + dflt} " + " ".join(tg)
            ln = f"L{lines[start]:<5}" if start in lines else "      "
            out(f"    {ln} {start:5d}: {nm} {arg}")
        for e in exc:
            out(f"    exc [{e[0]},{e[1]}) -> {e[2]} {cf.cname(e[3]) if e[3] else 'any'}")


def main():
    path = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else None
    cf = ClassFile(open(path, "rb").read())
    out = print
    out(f"class {cf.this} extends {cf.super} implements {cf.interfaces} (major {cf.major})")
    for acc, name, desc, attrs in cf.fields:
        cv = ""
        for an, d in attrs:
            if an == "ConstantValue":
                cv = " = " + cf.const(struct.unpack(">H", d)[0])
        out(f"  field {name}:{desc} acc=0x{acc:x}{cv}")
    for m in cf.methods:
        if filt and filt not in m[1]:
            continue
        dis_method(cf, m, out)


if __name__ == "__main__":
    main()
