"""Populate the reference's parameter objects from its own Jar/config.xml, the way its JAXB unmarshaller would.

Test infrastructure (build container only, used by tools/make_ref_exec.py).  The image has no JAXB runtime that could run
under tools/jvm_exec.py without the JDK, so this reads the classes' own javax.xml.bind annotations (XmlElement names,
XmlAccessType.FIELD) from the class files and assigns the values of the XML elements to the fields; constructors, defaults
and validate() are the reference's bytecode.  Nothing algorithmic lives here.
"""
import xml.etree.ElementTree as ET

from classdis import Reader
from jvm_exec import JBox, JObject, f32


def _annotations(cf, attrs):
    out = {}
    for an, data in attrs:
        if an != "RuntimeVisibleAnnotations":
            continue
        r = Reader(data)

        def elem():
            tag = chr(r.u1())
            if tag in "BCDFIJSZs":
                return cf.cp[r.u2()][1]
            if tag == "e":
                return (cf.utf(r.u2()), cf.utf(r.u2()))
            if tag == "c":
                return cf.utf(r.u2())
            if tag == "@":
                return ann()
            if tag == "[":
                return [elem() for _ in range(r.u2())]
            raise ValueError(tag)

        def ann():
            t = cf.utf(r.u2())
            kv = {}
            for _ in range(r.u2()):
                k = cf.utf(r.u2())
                kv[k] = elem()
            return t, kv

        for _ in range(r.u2()):
            t, kv = ann()
            out[t] = kv
    return out


def xml_fields(jvm, cname):
    """[(xml element name, field name, descriptor)] of a class and its superclasses (XmlAccessType.FIELD)"""
    res = []
    n = cname
    while n is not None and jvm.has_class(n):
        jc = jvm.load(n)
        for acc, fname, fdesc, attrs in jc.cf.fields:
            if acc & 0x0008 or acc & 0x0080:  # static, transient
                continue
            a = _annotations(jc.cf, attrs)
            if "Ljavax/xml/bind/annotation/XmlTransient;" in a:
                continue
            xe = a.get("Ljavax/xml/bind/annotation/XmlElement;", {})
            name = xe.get("name", fname)
            res.append((fname if name == "##default" else name, fname, fdesc))
        n = jc.super
    return res


def _convert(jvm, desc, text, elem):
    t = (text or "").strip()
    if desc == "Ljava/lang/Integer;":
        return JBox("java/lang/Integer", int(t))
    if desc == "Ljava/lang/Long;":
        return JBox("java/lang/Long", int(t))
    if desc == "Ljava/lang/Float;":
        return JBox("java/lang/Float", f32(float(t)))
    if desc == "Ljava/lang/Double;":
        return JBox("java/lang/Double", float(t))
    if desc == "Ljava/lang/Boolean;":
        return JBox("java/lang/Boolean", 1 if t in ("true", "1") else 0)
    if desc in ("I", "S", "B", "J"):
        return int(t)
    if desc == "Z":
        return 1 if t in ("true", "1") else 0
    if desc == "F":
        return f32(float(t))
    if desc == "D":
        return float(t)
    if desc == "Ljava/lang/String;":
        return text if text is not None else ""
    cname = desc[1:-1]
    if jvm.has_class(cname):
        jc = jvm.init_class(cname)
        if jc.is_enum:
            return jvm.call_static(cname, "valueOf", f"(Ljava/lang/String;)L{cname};", t)
        return unmarshal(jvm, cname, elem)
    raise ValueError(f"no conversion for {desc}")


def unmarshal(jvm, cname, elem, report=None):
    obj = jvm.new(cname)
    fields = {x: (f, d) for x, f, d in xml_fields(jvm, cname)}
    for child in elem:
        if child.tag not in fields:
            if report is not None:
                report.append(f"{cname}: no field for <{child.tag}>")
            continue
        fname, fdesc = fields[child.tag]
        try:
            obj.f[fname] = _convert(jvm, fdesc, child.text, child)
        except ValueError:
            # e.g. <mergeBCsED>null</mergeBCsED> into an Integer: JAXB's default ValidationEventHandler reports the conversion
            # error and goes on, the field keeps the value the constructor gave it
            if report is not None:
                report.append(f"{cname}.{fname}: unparsable {child.text!r}, left at its default")
    return obj


def load_config(jvm, cname, path="/root/reference/Jar/config.xml"):
    report = []
    root = ET.parse(path).getroot()
    return unmarshal(jvm, cname, root, report), report
