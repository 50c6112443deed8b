#!/usr/bin/env python3
"""Symbolic-stack folder on top of classdis.py: one pseudo-statement per store / call / branch.

Test infrastructure only (see classdis.py).  Output is reference content: scratch dirs only.

usage: classfold.py Foo.class [method-substring]
"""
import re
import struct
import sys

from classdis import OPS, ATYPE, ClassFile, Reader

BIN = {
    "add": "+", "sub": "-", "mul": "*", "div": "/", "rem": "%", "shl": "<<", "shr": ">>", "ushr": ">>>",
    "and": "&", "or": "|", "xor": "^",
}
CMP = {"eq": "==", "ne": "!=", "lt": "<", "ge": ">=", "gt": ">", "le": "<="}


def parse_desc(desc):
    """-> (list of arg cats, return cat (0 = void))"""
    args = []
    i = 1
    while desc[i] != ")":
        c = desc[i]
        cat = 2 if c in "JD" else 1
        while desc[i] == "[":
            i += 1
            cat = 1
        if desc[i] == "L":
            i = desc.index(";", i)
        i += 1
        args.append(cat)
    r = desc[i + 1 :]
    rc = 0 if r == "V" else (2 if r in ("J", "D") else 1)
    return args, rc


def short(cls):
    return cls.split("/")[-1]


def fold_method(cf, m, out):
    acc, name, desc, attrs = m
    out(f"\n== {name}{desc} acc=0x{acc:x}")
    for an, data in attrs:
        if an != "Code":
            continue
        r = Reader(data)
        r.u2(), r.u2()
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

        def lv(slot, pc, store=False, width=1):
            for spc, ln, nm, ds, sl in lvt:
                if sl == slot and (spc <= pc < spc + ln or (store and spc - 5 <= pc < spc + ln)):
                    return nm
            return f"v{slot}"

        # decode instructions first
        ins = []
        pc = 0
        n = len(code)
        while pc < n:
            op = code[pc]
            nm, kind = OPS.get(op, (f"op{op}", ""))
            start = pc
            pc += 1
            a = None
            if kind == "s1":
                a = struct.unpack_from(">b", code, pc)[0]; pc += 1
            elif kind == "s2":
                a = struct.unpack_from(">h", code, pc)[0]; pc += 2
            elif kind == "u1":
                a = code[pc]; pc += 1
            elif kind == "cp1":
                a = code[pc]; pc += 1
            elif kind == "cp2":
                a = struct.unpack_from(">H", code, pc)[0]; pc += 2
            elif kind == "lv1":
                a = code[pc]; pc += 1
            elif kind.startswith("lvimp"):
                a = int(kind[-1]); nm = nm[:-2]
            elif kind == "iinc":
                a = (code[pc], struct.unpack_from(">b", code, pc + 1)[0]); pc += 2
            elif kind == "br2":
                a = start + struct.unpack_from(">h", code, pc)[0]; pc += 2
            elif kind == "br4":
                a = start + struct.unpack_from(">i", code, pc)[0]; pc += 4
            elif kind in ("ii", "id"):
                a = struct.unpack_from(">H", code, pc)[0]; pc += 4
            elif kind == "mana":
                a = (struct.unpack_from(">H", code, pc)[0], code[pc + 2]); pc += 3
            elif kind == "wide":
                op2 = code[pc]
                nm = OPS[op2][0]
                idx = struct.unpack_from(">H", code, pc + 1)[0]
                if op2 == 132:
                    a = (idx, struct.unpack_from(">h", code, pc + 3)[0]); pc += 5
                else:
                    a = idx; pc += 3
            elif kind == "tsw":
                pc = (pc + 3) & ~3
                dflt, lo, hi = struct.unpack_from(">iii", code, pc); pc += 12
                tg = []
                for k in range(hi - lo + 1):
                    tg.append((lo + k, start + struct.unpack_from(">i", code, pc)[0])); pc += 4
                a = (start + dflt, tg)
            elif kind == "lsw":
                pc = (pc + 3) & ~3
                dflt, np_ = struct.unpack_from(">ii", code, pc); pc += 8
                tg = []
                for k in range(np_):
                    mv, off = struct.unpack_from(">ii", code, pc); pc += 8
                    tg.append((mv, start + off))
                a = (start + dflt, tg)
            ins.append((start, nm, a))
        targets = {}
        for start, nm, a in ins:
            if nm.startswith("if") or nm in ("goto", "goto_w", "jsr", "jsr_w"):
                targets.setdefault(a, None)
            elif nm in ("tableswitch", "lookupswitch"):
                targets.setdefault(a[0], None)
                for _, t in a[1]:
                    targets.setdefault(t, None)
        handlers = {e[2]: (cf.cname(e[3]) if e[3] else "any") for e in exc}

        stack = []  # (expr, cat)
        curline = [None]

        def emit(pc, s):
            ln = lines.get(pc)
            # find most recent line
            if ln is None:
                ln = curline[0]
            out(f"  L{ln if ln is not None else '?':<4} {pc:5d}: {s}")

        def push(e, cat=1):
            stack.append((e, cat))

        def pop():
            if not stack:
                return "<?>"
            return stack.pop()[0]

        def spill(pc):
            # flush the stack to depth-named temps
            for d, (e, cat) in enumerate(stack):
                nm_ = f"$s{d}"
                if e != nm_:
                    emit(pc, f"{nm_} = {e}")
                    stack[d] = (nm_, cat)

        def paren(e):
            return e if re.fullmatch(r"[\w$.\[\]<>]+(\(.*\))?", e) else f"({e})"

        dead = False
        for idx, (pc, nm, a) in enumerate(ins):
            if pc in lines:
                curline[0] = lines[pc]
            if pc in targets or pc in handlers:
                if not dead:
                    spill(pc)
                    depth = len(stack)
                    if targets.get(pc) is None:
                        targets[pc] = [c for _, c in stack]
                cats = targets.get(pc)
                if pc in handlers:
                    stack[:] = [(f"<caught {short(handlers[pc])}>", 1)]
                elif cats is not None:
                    stack[:] = [(f"$s{d}", c) for d, c in enumerate(cats)]
                elif dead:
                    stack[:] = []
                out(f"  label_{pc}:")
                dead = False
            t = nm[0]
            if nm == "nop":
                pass
            elif nm == "aconst_null":
                push("null")
            elif nm.startswith("iconst_"):
                push("-1" if nm.endswith("m1") else nm[-1])
            elif nm.startswith("lconst_"):
                push(nm[-1] + "L", 2)
            elif nm.startswith("fconst_"):
                push(nm[-1] + ".0f")
            elif nm.startswith("dconst_"):
                push(nm[-1] + ".0", 2)
            elif nm in ("bipush", "sipush"):
                push(str(a))
            elif nm in ("ldc", "ldc_w", "ldc2_w"):
                e = cf.cp[a]
                if e[0] == "String":
                    push(repr(cf.utf(e[1])).replace("'", '"'))
                elif e[0] == "Class":
                    push(short(cf.utf(e[1])) + ".class")
                elif e[0] == "Long":
                    push(f"{e[1]}L", 2)
                elif e[0] == "Double":
                    push(f"{e[1]!r}d", 2)
                elif e[0] == "Float":
                    push(f"{e[1]!r}f")
                else:
                    push(str(e[1]))
            elif nm in ("iload", "lload", "fload", "dload", "aload"):
                push(lv(a, pc), 2 if t in "ld" else 1)
            elif nm in ("istore", "lstore", "fstore", "dstore", "astore"):
                emit(pc, f"{lv(a, pc, True)} = {pop()}")
            elif nm.endswith("aload") and len(nm) == 6:
                i_ = pop(); arr = pop()
                push(f"{paren(arr)}[{i_}]", 2 if t in "ld" else 1)
            elif nm.endswith("astore") and len(nm) == 7:
                v = pop(); i_ = pop(); arr = pop()
                emit(pc, f"{paren(arr)}[{i_}] = {v}")
            elif nm == "pop":
                e = pop()
                if "(" in e:
                    emit(pc, e)
            elif nm == "pop2":
                e, c = stack.pop()
                if c == 1:
                    stack.pop()
                if "(" in e:
                    emit(pc, e)
            elif nm == "dup":
                e, c = stack[-1]
                if e.startswith("new ") and "(" not in e:
                    stack.append((e, c))
                else:
                    if not re.fullmatch(r"[\w$.]+", e):
                        tn = f"$d{pc}"
                        emit(pc, f"{tn} = {e}")
                        stack[-1] = (tn, c)
                        e = tn
                    stack.append((e, c))
            elif nm == "dup_x1":
                a1 = stack.pop(); a2 = stack.pop()
                if not re.fullmatch(r"[\w$.]+", a1[0]):
                    tn = f"$d{pc}"; emit(pc, f"{tn} = {a1[0]}"); a1 = (tn, a1[1])
                stack.extend([a1, a2, a1])
            elif nm == "dup_x2":
                a1 = stack.pop(); a2 = stack.pop()
                if a2[1] == 2:
                    stack.extend([a1, a2, a1])
                else:
                    a3 = stack.pop(); stack.extend([a1, a3, a2, a1])
            elif nm == "dup2":
                a1 = stack[-1]
                if a1[1] == 2:
                    if not re.fullmatch(r"[\w$.]+", a1[0]):
                        tn = f"$d{pc}"; emit(pc, f"{tn} = {a1[0]}"); a1 = (tn, 2); stack[-1] = a1
                    stack.append(a1)
                else:
                    a2 = stack[-2]
                    for k, ent in ((-1, a1), (-2, a2)):
                        if not re.fullmatch(r"[\w$.\[\]]+", ent[0]):
                            tn = f"$d{pc}_{-k}"; emit(pc, f"{tn} = {ent[0]}"); stack[k] = (tn, ent[1])
                    stack.extend([stack[-2], stack[-1]])
            elif nm in ("dup2_x1", "dup2_x2"):
                a1 = stack.pop()
                if a1[1] == 2:
                    a2 = stack.pop()
                    if nm == "dup2_x2" and a2[1] == 1:
                        a3 = stack.pop(); stack.extend([a1, a3, a2, a1])
                    else:
                        stack.extend([a1, a2, a1])
                else:
                    a2 = stack.pop(); a3 = stack.pop()
                    stack.extend([a2, a1, a3, a2, a1])
            elif nm == "swap":
                a1 = stack.pop(); a2 = stack.pop(); stack.extend([a1, a2])
            elif nm[1:] in BIN and t in "ilfd":
                b = pop(); a_ = pop()
                push(f"({a_} {BIN[nm[1:]]} {b})", 2 if t in "ld" else 1)
            elif nm[1:] == "neg":
                push(f"(-{pop()})", 2 if t in "ld" else 1)
            elif nm == "iinc":
                emit(pc, f"{lv(a[0], pc)} += {a[1]}")
            elif re.fullmatch(r"[ilfd]2[ilfdbcs]", nm):
                tt = {"i": "int", "l": "long", "f": "float", "d": "double", "b": "byte", "c": "char", "s": "short"}[nm[2]]
                push(f"({tt}){paren(pop())}", 2 if nm[2] in "ld" else 1)
            elif nm in ("lcmp", "fcmpl", "fcmpg", "dcmpl", "dcmpg"):
                b = pop(); a_ = pop()
                push(f"{nm}({a_}, {b})")
            elif nm.startswith("if_icmp") or nm.startswith("if_acmp"):
                b = pop(); a_ = pop()
                spill(pc)
                emit(pc, f"if ({a_} {CMP[nm[-2:]]} {b}) goto label_{a}")
                if targets.get(a) is None:
                    targets[a] = [c for _, c in stack]
            elif nm in ("ifeq", "ifne", "iflt", "ifge", "ifgt", "ifle", "ifnull", "ifnonnull"):
                a_ = pop()
                spill(pc)
                if nm == "ifnull":
                    cond = f"{a_} == null"
                elif nm == "ifnonnull":
                    cond = f"{a_} != null"
                else:
                    mm = re.fullmatch(r"(lcmp|fcmpl|fcmpg|dcmpl|dcmpg)\((.*), (.*)\)", a_)
                    if mm and mm.group(2).count("(") == mm.group(2).count(")"):
                        cond = f"{mm.group(2)} {CMP[nm[2:]]} {mm.group(3)}" + ("" if mm.group(1) == "lcmp" else f" /*{mm.group(1)}*/")
                    else:
                        cond = f"{a_} {CMP[nm[2:]]} 0"
                emit(pc, f"if ({cond}) goto label_{a}")
                if targets.get(a) is None:
                    targets[a] = [c for _, c in stack]
            elif nm in ("goto", "goto_w"):
                spill(pc)
                emit(pc, f"goto label_{a}")
                if targets.get(a) is None:
                    targets[a] = [c for _, c in stack]
                dead = True
                stack[:] = []
            elif nm in ("tableswitch", "lookupswitch"):
                v = pop()
                spill(pc)
                emit(pc, f"switch ({v}) " + " ".join(f"{k}->label_{t_}" for k, t_ in a[1]) + f" default->label_{a[0]}")
                for _, t_ in a[1] + [(None, a[0])]:
                    if targets.get(t_) is None:
                        targets[t_] = [c for _, c in stack]
                dead = True
                stack[:] = []
            elif nm.endswith("return"):
                emit(pc, "return" + ("" if nm == "return" else " " + pop()))
                dead = True
                stack[:] = []
            elif nm == "athrow":
                emit(pc, f"throw {pop()}")
                dead = True
                stack[:] = []
            elif nm in ("getstatic", "getfield", "putstatic", "putfield"):
                e = cf.cp[a]
                nt = cf.cp[e[2]]
                fname, fdesc = cf.utf(nt[1]), cf.utf(nt[2])
                cls = short(cf.cname(e[1]))
                cat = 2 if fdesc in ("J", "D") else 1
                if nm == "getstatic":
                    push(f"{cls}.{fname}", cat)
                elif nm == "getfield":
                    push(f"{paren(pop())}.{fname}", cat)
                elif nm == "putstatic":
                    emit(pc, f"{cls}.{fname} = {pop()}")
                else:
                    v = pop(); o = pop()
                    emit(pc, f"{paren(o)}.{fname} = {v}")
            elif nm in ("invokevirtual", "invokespecial", "invokestatic", "invokeinterface"):
                e = cf.cp[a]
                nt = cf.cp[e[2]]
                mname, mdesc = cf.utf(nt[1]), cf.utf(nt[2])
                cls = short(cf.cname(e[1]))
                acats, rc = parse_desc(mdesc)
                args = [pop() for _ in acats][::-1]
                if nm == "invokestatic":
                    call = f"{cls}.{mname}({', '.join(args)})"
                    if rc:
                        push(call, rc)
                    else:
                        emit(pc, call)
                else:
                    o = pop()
                    if mname == "<init>":
                        if o.startswith("new ") and "(" not in o:
                            call = f"{o}({', '.join(args)})"
                            # replace duplicate on stack
                            if stack and stack[-1][0] == o:
                                stack[-1] = (call, 1)
                            else:
                                emit(pc, call)
                        else:
                            emit(pc, f"{o}.<init>[{cls}]({', '.join(args)})")
                    else:
                        sp = f"super[{cls}]." if nm == "invokespecial" and o == "this" else f"{paren(o)}."
                        call = f"{sp}{mname}({', '.join(args)})"
                        if rc:
                            push(call, rc)
                        else:
                            emit(pc, call)
            elif nm == "invokedynamic":
                e = cf.cp[a]
                nt = cf.cp[e[2]]
                mname, mdesc = cf.utf(nt[1]), cf.utf(nt[2])
                acats, rc = parse_desc(mdesc)
                args = [pop() for _ in acats][::-1]
                bs = cf.bootstrap[e[1]]
                bsm = cf.const(bs[0])
                if "makeConcatWithConstants" in bsm:
                    recipe = cf.const(bs[1][0])
                    call = f"concat({recipe}; {', '.join(args)})"
                else:
                    tgt = [cf.const(x) for x in bs[1]]
                    impl = [x for x in tgt if x.startswith("MH")]
                    impl = impl[0].split("] ")[1] if impl else str(tgt)
                    impl = impl.split(":")[0].split("/")[-1]
                    call = f"lambda[{mname} -> {impl}]({', '.join(args)})"
                if rc:
                    push(call, rc)
                else:
                    emit(pc, call)
            elif nm == "new":
                push("new " + short(cf.cname(a)))
            elif nm == "newarray":
                push(f"new {ATYPE.get(a, a)}[{pop()}]")
            elif nm == "anewarray":
                push(f"new {short(cf.cname(a))}[{pop()}]")
            elif nm == "multianewarray":
                dims = [pop() for _ in range(a[1])][::-1]
                push(f"new {cf.cname(a[0])}{''.join(f'[{d}]' for d in dims)}")
            elif nm == "arraylength":
                push(f"{paren(pop())}.length")
            elif nm == "checkcast":
                push(f"({short(cf.cname(a))}){paren(pop())}")
            elif nm == "instanceof":
                push(f"({pop()} instanceof {short(cf.cname(a))})")
            elif nm in ("monitorenter", "monitorexit"):
                emit(pc, f"{nm}({pop()})")
            else:
                emit(pc, f"??? {nm} {a}")
        for e in exc:
            out(f"  exc [{e[0]},{e[1]}) -> label_{e[2]} {cf.cname(e[3]) if e[3] else 'any'}")


def main():
    path = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else None
    cf = ClassFile(open(path, "rb").read())
    print(f"class {cf.this} extends {cf.super} implements {cf.interfaces}")
    for acc, name, desc, attrs in cf.fields:
        cv = ""
        for an, d in attrs:
            if an == "ConstantValue":
                cv = " = " + cf.const(struct.unpack(">H", d)[0])
        print(f"  field {name}:{desc} acc=0x{acc:x}{cv}")
    for m in cf.methods:
        if filt and filt not in m[1]:
            continue
        try:
            fold_method(cf, m, print)
        except Exception as ex:  # fall back to raw listing on a folding failure
            print(f"  !! fold failed: {ex!r}")


if __name__ == "__main__":
    main()
