"""CPU checks of the drop-in boundary: the library builds for gfx950, loads, exports every symbol that
include/sicelore_mi.h declares, the struct layouts agree, and without a GPU it fails loudly (no fallback)."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

import __graft_entry__ as graft

HEADER = os.path.join(graft.ROOT, "include", "sicelore_mi.h")


@pytest.fixture(scope="module")
def built():
    graft.build()
    return graft.load_package()


def _declared_symbols():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(smi_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(built):
    lib = built.load_library()
    declared = _declared_symbols()
    assert len(declared) >= 10
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in sicelore_mi.h but not exported"
    from sicelore_amd import lib as libmod

    assert sorted(libmod.EXPORTS) == declared


def test_code_object_targets_gfx950(built):
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", built.library_path()],
                         capture_output=True, text=True).stdout
    assert "gfx950" in out


def test_struct_layout_matches_header(built, tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "sicelore_mi.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   "sizeof(smi_bc_window),offsetof(smi_bc_window,nmask),offsetof(smi_bc_window,flags),"
                   "sizeof(smi_bc_result),offsetof(smi_bc_result,ed_sec),offsetof(smi_bc_result,found),"
                   "offsetof(smi_bc_result,ins_minus_del),offsetof(smi_bc_result,n_matches));return 0;}\n")
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.dirname(HEADER), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    w, r = built.BC_WINDOW_DTYPE, built.BC_RESULT_DTYPE
    assert got == [w.itemsize, w.fields["nmask"][1], w.fields["flags"][1], r.itemsize, r.fields["ed_sec"][1],
                   r.fields["found"][1], r.fields["ins_minus_del"][1], r.fields["n_matches"][1]]


def test_struct_layout_of_the_other_records(built, tmp_path):
    from sicelore_amd import lib as libmod

    src = tmp_path / "sz2.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "sicelore_mi.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   "sizeof(smi_scan_result),offsetof(smi_scan_result,scan_end),offsetof(smi_scan_result,found),"
                   "offsetof(smi_scan_result,tso_end),sizeof(smi_scan_config),offsetof(smi_scan_config,adapter4),"
                   "offsetof(smi_scan_config,adapter_search_window),sizeof(smi_chimera_result),offsetof(smi_chimera_result,flags),"
                   "sizeof(smi_umi_assignment),offsetof(smi_umi_assignment,pos2),sizeof(smi_umi_cluster_config));return 0;}\n")
    exe = tmp_path / "sz2"
    subprocess.check_call(["gcc", "-I", os.path.dirname(HEADER), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    sr, sc, cr = built.SCAN_RESULT_DTYPE, built.SCAN_CONFIG_DTYPE, built.CHIMERA_RESULT_DTYPE
    ua, uc = libmod.UMI_ASSIGNMENT_DTYPE, libmod.UMI_CLUSTER_CONFIG_DTYPE
    assert got == [sr.itemsize, sr.fields["scan_end"][1], sr.fields["found"][1], sr.fields["tso_end"][1], sc.itemsize,
                   sc.fields["adapter4"][1], sc.fields["adapter_search_window"][1], cr.itemsize, cr.fields["flags"][1],
                   ua.itemsize, ua.fields["pos2"][1], uc.itemsize]


def test_host_entry_points_reject_bad_arguments(built):
    """host-side entry points need no GPU: they validate and report through smi_last_error"""
    from sicelore_amd import lib as libmod

    lib = built.load_library()
    assert lib.smi_region_group(None, None, None, 5, 500, 0, None, None) < 0
    assert b"smi_region_group" in lib.smi_last_error()
    assert lib.smi_umi_cluster_groups(None, None, None, 3, None, None, None, None, 1) < 0
    assert lib.smi_chimera_fragment_name(b"r x", None, 0, None, 0) < 0
    assert libmod.region_group([], []) == ([], 0)
    out, sk = libmod.umi_cluster_groups(np.zeros(0, np.uint8), [0], [0], np.zeros(0, np.float32))
    assert out.size == 0 and sk.size == 0


def test_no_gpu_fails_loudly(built):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(built.SmiError, match="no HIP device|hip"):
        built.Context(0)


def test_missing_library_is_an_error(built, monkeypatch):
    from sicelore_amd import lib as libmod

    monkeypatch.setattr(libmod, "_LIB", None)
    monkeypatch.setattr(libmod, "library_path", lambda: "/nonexistent/libsicelore_mi.so")
    with pytest.raises(built.SmiError, match="missing"):
        libmod.load_library()


def test_synth_generator_is_seeded(synth):
    wl = synth.make_whitelist(5000, seed=5)
    assert wl.numel() == 5000 and int(wl.unique().numel()) == 5000
    assert (synth.make_whitelist(5000, seed=5) == wl).all()
    used = synth.pick_used(wl, 50, seed=6)
    a = synth.gen_bc_region(1000, used, seed=7)
    b = synth.gen_bc_region(1000, used, seed=7)
    assert (a["codes"] == b["codes"]).all() and (a["ae"] == b["ae"]).all()
    w = synth.pack_windows(a["codes"], a["ae"])
    assert w.shape == (1000, 2) and bool(((w[:, 1] >> 32) & 1).all())


def test_struct_layouts_of_the_host_side_units(pkg, tmp_path):
    """records of the BAM index, the UMI tags and the configs of the chunk workers: numpy / ctypes views == the C header"""
    import ctypes

    from sicelore_amd import lib as libmod

    src = tmp_path / "t2.c"
    src.write_text("#include <stdio.h>\n#include <stddef.h>\n#include \"sicelore_mi.h\"\nint main(){printf(\"%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n\","
                   "sizeof(smi_bam_record),offsetof(smi_bam_record,rec_len),offsetof(smi_bam_record,flag),offsetof(smi_bam_record,l_read_name),"
                   "sizeof(smi_umi_tag),offsetof(smi_umi_tag,flags),offsetof(smi_umi_tag,u7),"
                   "sizeof(smi_pass2_config),offsetof(smi_pass2_config,rank_keys),sizeof(smi_pass2_output),offsetof(smi_pass2_output,fastq_errors),"
                   "sizeof(smi_assignumis_config));return 0;}\n")
    exe = tmp_path / "t2"
    subprocess.check_call(["gcc", "-I", os.path.dirname(HEADER), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    br, ut = libmod.BAM_RECORD_DTYPE, libmod.UMI_TAG_DTYPE
    assert got == [br.itemsize, br.fields["rec_len"][1], br.fields["flag"][1], br.fields["l_read_name"][1], ut.itemsize,
                   ut.fields["flags"][1], ut.fields["u7"][1], ctypes.sizeof(libmod.Pass2Config), libmod.Pass2Config.rank_keys.offset,
                   ctypes.sizeof(libmod.Pass2Output), libmod.Pass2Output.fastq_errors.offset, ctypes.sizeof(libmod.AssignUmisConfig)]


def test_size_t_return_is_not_truncated(built):
    """smi_read_planes_words returns size_t: 2^36 bases need more words than a C int holds"""
    from sicelore_amd import lib as libmod

    lib = built.load_library()
    assert lib.smi_read_planes_words.restype is ctypes.c_size_t
    small = lib.smi_read_planes_words(3200, 1)
    assert small == 4 * (3200 // 32 + 5) or small > 0
    big = lib.smi_read_planes_words(1 << 36, 1)
    assert big >= (1 << 36) // 8 and big > 2 ** 31
    assert libmod.EXPORTS.count("smi_read_planes_words") == 1
