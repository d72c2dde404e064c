"""CPU-side checks of the drop-in boundary: the shared library loads, exports every symbol that
include/vadc_amd.h declares (and nothing is declared that is not exported), and fails loudly -- not with a
CPU fallback -- when there is no GPU.  No compute calls here."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

from conftest import ROOT
from vadc_amd import _lib


def header_symbols():
    text = open(os.path.join(ROOT, "include", "vadc_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vadc_amd_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_exported():
    L = _lib.load()
    syms = header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/vadc_amd.h but not exported by libvadc_amd.so"
    assert sorted(_lib.SYMBOLS) == syms


def test_abi_revision_is_checked_before_a_struct_crosses_the_boundary():
    """vadc_amd_get_caps writes sizeof(vadc_amd_caps) of the LIBRARY's revision: a binding compiled against another revision must find out first.  The header's
    VADC_AMD_ABI_VERSION, the library's vadc_amd_abi_version() and the ctypes mirror's ABI_VERSION agree; the mirror's Caps has the header's fields in its order;
    the adapter header checks the revision in backend_init and asks for its own sizeof only"""
    text = open(os.path.join(ROOT, "include", "vadc_amd.h")).read()
    rev = int(re.search(r"#define\s+VADC_AMD_ABI_VERSION\s+(\d+)", text).group(1))
    L = _lib.load()
    assert L.vadc_amd_abi_version() == rev == _lib.ABI_VERSION
    body = re.search(r"typedef struct vadc_amd_caps \{(.*?)\} vadc_amd_caps;", text, re.S).group(1)
    fields = re.findall(r"int32_t\s+([a-z0-9_]+);", re.sub(r"/\*.*?\*/", "", body, flags=re.S))
    assert fields == [n for n, _ in _lib.Caps._fields_]
    adapter = open(os.path.join(ROOT, "include", "vadc_backend_hip.h")).read()
    assert "vadc_amd_abi_version() != VADC_AMD_ABI_VERSION" in adapter and "vadc_amd_get_caps_sized(engine, &caps, sizeof caps)" in adapter


def test_exports_are_plain_c_and_only_ours():
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = sorted(l.split()[-1] for l in out.splitlines() if " T " in l)
    ours = [s for s in exported if s.startswith("vadc_amd_")]
    assert sorted(ours) == header_symbols()
    # no torch / C++-mangled API surface at the boundary
    assert not [s for s in exported if "torch" in s or "at::" in s]


def test_library_carries_gfx950_code_object():
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    for other in (b"gfx90a", b"gfx942", b"sm_80"):
        assert other not in blob


def test_create_fails_loudly_without_gpu(weights_blob):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from vadc_amd.engine import Engine, VadcAmdError
    with pytest.raises(VadcAmdError) as ei:
        Engine(weights_blob)
    assert ei.value.code == -3 and "no CPU fallback" in str(ei.value)


def test_bad_arguments_rejected_before_touching_the_device(weights_blob):
    L = _lib.load()
    h = C.c_void_p()
    assert L.vadc_amd_create(None, 0, -1, 1, 1, 0, C.byref(h)) == -2
    assert L.vadc_amd_create(weights_blob, len(weights_blob), -1, 0, 1, 0, C.byref(h)) == -1
    assert L.vadc_amd_create(weights_blob, len(weights_blob), -1, 1, 1, 7, C.byref(h)) == -1
    assert L.vadc_amd_create(weights_blob[:-8], len(weights_blob) - 8, -1, 1, 1, 0, C.byref(h)) == -2
    assert b"testtensor" in L.vadc_amd_last_error()
    assert L.vadc_amd_run_f32(None, None, 1, 1, None) == -1
    assert L.vadc_amd_kernel_name(0) == b"k_frontend"


def test_product_never_imports_the_oracle():
    """The shipped path must not route through oracle/ (or any CPU implementation)."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "vadc_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".c")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text.lower().replace("no cpu fallback", ""), os.path.join(dirpath, f)
    out = subprocess.check_output(["ldd", _lib.LIB_PATH], text=True)
    assert "oracle" not in out and "libvadc_ref" not in out


def test_adapter_header_compiles_against_the_reference_vadc_h():
    """include/vadc_backend_hip.h (the literal backend_init / backend_create_tensors / backend_run trio) inside a translation unit that includes the
    REFERENCE's own vadc.h: the types it touches (MemoryArena, String8, Silero_Config, VADC_Context, Tensor_Buffers) are the reference's.  Build
    container only: /root/reference does not travel."""
    ref = "/root/reference"
    if not os.path.exists(os.path.join(ref, "vadc.h")):
        pytest.skip("reference tree not present")
    src = os.path.join(ROOT, "tests", "c", "adapter_check.c")
    r = subprocess.run(["gcc", "-std=gnu11", "-Wall", "-fsyntax-only", "-include", "stddef.h", "-DONNX_INFERENCE_ENABLED=0",
                        "-I", ref, "-I", os.path.join(ROOT, "include"), src], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_layout_mirrors_match_the_reference_headers():
    """tests/c/vadc_layout_mirror.h (what tests/c/adapter_run.c drives the backend trio with on the GPU box) against the reference's vadc.h / string8.h:
    sizeof and every field's offset and width, by _Static_assert (tests/c/layout_check.c).  Build container only."""
    ref = "/root/reference"
    if not os.path.exists(os.path.join(ref, "vadc.h")):
        pytest.skip("reference tree not present")
    r = subprocess.run(["gcc", "-std=gnu11", "-fsyntax-only", "-include", "stddef.h", "-DONNX_INFERENCE_ENABLED=0", "-I", ref,
                        "-I", os.path.join(ROOT, "tests", "c"), os.path.join(ROOT, "tests", "c", "layout_check.c")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_adapter_runner_compiles_over_the_mirrors():
    """tests/c/adapter_run.c = the backend trio driven like vadc.c:686-795 / 56-162 over the layout mirrors: must compile wherever gcc is (it RUNS in
    tests/test_gpu_adapter.py)"""
    r = subprocess.run(["gcc", "-std=gnu11", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "tests", "c"),
                        os.path.join(ROOT, "tests", "c", "adapter_run.c")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_multi_gpu_c_host_compiles():
    """host/vadc_hip_multi.c (one engine per GPU, ncclGather of the probabilities: the north star's multi-GPU host in C) must compile as C against the
    HIP runtime's and RCCL's C APIs wherever the ROCm headers are (it RUNS in tests/test_gpu_multi_host.py)"""
    if not os.path.exists("/opt/rocm/include/rccl/rccl.h"):
        pytest.skip("no ROCm headers")
    r = subprocess.run(["gcc", "-std=gnu11", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
                        os.path.join(ROOT, "host", "vadc_hip_multi.c")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_layer1_kernel_layout_invariants():
    """what k_layer1_regs relies on, as static_asserts over the header the kernel and the engine's image packer share (tests/c/l1_layout_check.cpp): the
    channel -> operand-slot map is a bijection and bank-conflict free, the DMA groups overwrite consumed bytes only, image + wave buffers fit the LDS"""
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I", os.path.join(ROOT, "vadc_amd", "csrc"), os.path.join(ROOT, "tests", "c", "l1_layout_check.cpp")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_counted_wait_kernels_do_not_spill(tmp_path):
    """k_layer1_regs / k_layer1_regs_v4 / k_lstm_layer wait for their LDS-DMA pieces with COUNTED s_waitcnt vmcnt(N): the count assumes that the only
    vector-memory operations a wave issues per iteration are the ones the source lists, and an in-flight load's destination register must not be moved.  A
    register spill (scratch traffic is vector-memory traffic), another operation count or a compiler-made write of M0 would break that silently -- the same
    checker the Makefile runs on every build (tools/check_counted_waits.py), here on fresh listings; and the checker itself must catch a doctored listing."""
    import shutil
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_counted_waits as ccw
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src_dir = os.path.join(ROOT, "vadc_amd", "csrc")
    outs = []
    for src in ("kernels_layer1_regs.hip", "kernels_layer1_regs_v4.hip", "kernels_lstm.hip"):
        out = str(tmp_path / (src + ".s"))
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only", "-Wno-unused-command-line-argument",
                            "-I", os.path.join(ROOT, "include"), "-o", out, os.path.join(src_dir, src)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(out)
    seen = set()
    for o in outs:
        errors, s_ = ccw.check(o)
        assert not errors, errors
        seen |= s_
    assert seen == set(ccw.RULES)                                   # every kernel the rules name was found in a listing
    # a listing with one DMA piece missing, and one with a spill inside the layer-1 kernel, must be rejected
    txt = open(outs[0]).read()
    bad1 = str(tmp_path / "missing_piece.s")
    i = txt.index("_ZN4vadc13k_layer1_regsILi12ELi0")
    j = txt.index("global_load_lds_dwordx4", txt.index(":", i))
    open(bad1, "w").write(txt[:j] + "s_nop 0 ;" + txt[j + len("global_load_lds_dwordx4"):])
    assert ccw.check(bad1)[0]
    bad2 = str(tmp_path / "spill.s")
    open(bad2, "w").write(txt[:j] + "scratch_store_dword off, v1, off\n\tglobal_load_lds_dwordx4" + txt[j + len("global_load_lds_dwordx4"):])
    assert ccw.check(bad2)[0]


def test_no_packed_add_takes_its_low_result_from_a_later_sources_high_dword(tmp_path):
    """tools/check_pk_opsel.py (run by the Makefile on the listing of EVERY kernel file): `v_pk_add_f32 d, a, b op_sel:[0,1]` -- low result = a.lo + b.HI -- was measured
    to return a.lo alone in lanes 48 .. 63, rarely, beside another kernel's waves; k_frontend_ri writes the swapped pair as the first source
    instead, and hipcc's own packed horizontal sums (k_enc_fused, with SLP vectorisation) are kept out by -fno-slp-vectorize.  The checker on synthetic listings:
    it must pass the safe forms and catch the risky ones; and on a fresh listing of the front end, whose SRC1 template form must not be instantiated"""
    import shutil
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_pk_opsel as cpo
    ok = tmp_path / "ok.s"
    ok.write_text("_Z1kv:\n\tv_pk_add_f32 v[0:1], v[2:3], v[4:5]\n\tv_pk_add_f32 v[0:1], v[2:3], v[4:5] op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[1,0]\n"
                  "\tv_pk_mul_f32 v[0:1], v[2:3], s[4:5] op_sel_hi:[1,0]\n\tv_pk_fma_f32 v[0:1], v[2:3], v[4:5], -1.0 op_sel_hi:[1,1,0]\n\tv_pk_mov_b32 v[0:1], v[2:3], v[4:5] op_sel:[1,0]\n")
    assert cpo.check(str(ok)) == []
    for line in ("v_pk_add_f32 v[0:1], v[2:3], v[4:5] op_sel:[0,1] op_sel_hi:[1,0]", "v_pk_mul_f32 v[0:1], v[2:3], v[4:5] op_sel:[1,1]", "v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[0,0,1]"):
        bad = tmp_path / "bad.s"
        bad.write_text("_Z1kv:\n\t" + line + "\n")
        found = cpo.check(str(bad))
        assert len(found) == 1 and found[0][1] == "_Z1kv"
        assert cpo.main([str(bad)]) == 1
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    out = str(tmp_path / "fe.s")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-S", "--cuda-device-only", "-Wno-unused-command-line-argument",
                        "-o", out, os.path.join(ROOT, "vadc_amd", "csrc", "kernels_frontend.hip")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert cpo.check(out) == []
    assert "k_frontend_sym" in open(out).read()                     # the shipped form is in the listing that was checked

