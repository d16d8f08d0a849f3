"""The C-ABI library: loads, exports every symbol include/qinco_hip.h declares, struct layouts agree with the
ctypes mirror.  No compute calls here (no GPU in this tier)."""
import ctypes as C
import re
import subprocess

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def lib():
    from qinco_amd import _lib, build
    build.build()
    return _lib.load()


def declared_symbols():
    txt = (ROOT / "include" / "qinco_hip.h").read_text()
    return sorted(set(re.findall(r"QINCO_API[^;]*?\b(qinco_\w+)\s*\(", txt)))


def test_header_symbols_exported(lib):
    from qinco_amd import _lib
    syms = declared_symbols()
    assert len(syms) >= 14 and set(syms) == set(_lib.API_SYMBOLS)
    for s in syms:
        assert getattr(lib, s) is not None
    out = subprocess.run(["nm", "-D", "--defined-only", str(_lib.LIB_PATH)], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (\w+)", out))
    assert set(syms) <= exported
    assert all(e.startswith("qinco_") for e in exported), exported   # nothing else leaks out of the .so


def test_library_contains_gfx950_code_objects():
    from qinco_amd import _lib
    data = _lib.LIB_PATH.read_bytes()
    assert b"gfx950" in data and b"mlp_kernel" in data
    for other in (b"gfx942", b"gfx90a", b"sm_90"):
        assert other not in data


def test_struct_layouts_match_header():
    from qinco_amd import _lib
    src = r'''
    #include "qinco_hip.h"
    #include <stddef.h>
    #include <stdio.h>
    int main(void) {
      printf("%zu %zu %zu %zu %zu %zu\n", sizeof(qinco_desc), offsetof(qinco_desc, max_batch), offsetof(qinco_desc, qinco1_mode),
             sizeof(qinco_weights), offsetof(qinco_weights, codebook), offsetof(qinco_weights, down));
      return 0; }'''
    import tempfile, os
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", str(ROOT / "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")])
        vals = [int(v) for v in subprocess.check_output([os.path.join(d, "t")]).split()]
    D, W = _lib.QincoDesc, _lib.QincoWeights
    assert vals == [C.sizeof(D), D.max_batch.offset, D.qinco1_mode.offset, C.sizeof(W), W.codebook.offset, W.down.offset]


def test_shape_table(lib):
    assert lib.qinco_shape_supported(128, 384, 384) == 1      # C2 / C3
    assert lib.qinco_shape_supported(128, 128, 256) == 1      # C1
    assert lib.qinco_shape_supported(768, 384, 384) == 1      # C4
    assert lib.qinco_shape_supported(100, 384, 384) == 1      # zero-padded to the compiled-in (128, 384, 384)
    assert lib.qinco_shape_supported(128, 416, 384) == 0      # a shape nobody compiled (until ensure_instance builds it)
    assert lib.qinco_version().startswith(b"qinco_hip")


def test_create_ex_validates_its_flags_before_touching_the_device(lib):
    """qinco_create_ex: unknown flag bits are an argument error; the split-fp16 flag on a shape without a split instance (or a
    model without FFN blocks) is QINCO_ERR_UNSUPPORTED -- both decided on the host, no GPU needed."""
    from qinco_amd import QincoConfig, _lib, synth_state_dict
    from qinco_amd.engine import QincoEngine
    h = C.c_void_p()
    d = _lib.QincoDesc(D=128, De=384, Dh=384, L=16, M=2, K=256, A=16, B=8, qinco1_mode=0, ivf_K=0, max_batch=64)
    w = _lib.QincoWeights()
    w.data_std = 1.0
    assert lib.qinco_create_ex(C.byref(d), C.byref(w), 1 << 10, C.byref(h)) == -1
    assert b"unknown flag" in lib.qinco_last_error()
    d.Dh = 96                                   # (128, 384, 96) has no instance at all, let alone a split one
    assert lib.qinco_create_ex(C.byref(d), C.byref(w), _lib.CREATE_SPLIT_F16, C.byref(h)) == -3
    assert b"split-fp16" in lib.qinco_last_error()
    d.Dh, d.L = 384, 0                          # L = 0 runs as one all-zero FFN block: accepted, the next check (weights) fires
    assert lib.qinco_create_ex(C.byref(d), C.byref(w), _lib.CREATE_SPLIT_F16, C.byref(h)) == -1
    assert b"missing data_mean" in lib.qinco_last_error()
    # qinco_create_opt: the options struct carries its own size; a mismatch (an older / newer header) is an argument error
    o = _lib.QincoOptions(struct_bytes=C.sizeof(_lib.QincoOptions) - 8, create_flags=0, mlp_P=-1, mlp_var=-1, table_coop_max=-1)
    assert lib.qinco_create_opt(C.byref(d), C.byref(w), C.byref(o), C.byref(h)) == -1
    assert b"struct_bytes" in lib.qinco_last_error()
    o.struct_bytes = C.sizeof(_lib.QincoOptions)
    o.create_flags = 1 << 10
    assert lib.qinco_create_opt(C.byref(d), C.byref(w), C.byref(o), C.byref(h)) == -1
    assert b"unknown flag" in lib.qinco_last_error()
    cfg = QincoConfig(D=32, M=2, K=256, L=1, de=64, dh=96)          # odd number of hidden blocks: no split instance
    with pytest.raises(NotImplementedError, match="split-fp16"):
        QincoEngine(cfg, synth_state_dict(cfg, 1), split_f16=True)


def test_fails_loudly_without_gpu():
    """No CPU fallback: on a box without a GPU creating an engine must raise, not silently compute elsewhere."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from qinco_amd import QincoConfig, QincoEngine, synth_state_dict
    cfg = QincoConfig(D=32, M=2, K=256, L=1, de=None, dh=64)
    with pytest.raises(RuntimeError):
        QincoEngine(cfg, synth_state_dict(cfg, 1))


def test_missing_library_raises(monkeypatch, tmp_path):
    from qinco_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setenv("QINCO_HIP_LIB", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.QincoLibraryError, match="no CPU fallback"):
        _lib.load()


def test_product_never_imports_oracle():
    """oracle/ is test infrastructure: nothing under qinco_amd/ may import or reference it."""
    for p in (ROOT / "qinco_amd").rglob("*.py"):
        txt = p.read_text()
        assert "oracle" not in txt.replace("oracle-", ""), p


def test_kernel_instance_built_on_demand_and_registered(lib):
    """A geometry shapes.def does not list: qinco_padded_shape gives the 32-feature-block shape, ensure_instance compiles ONE
    translation unit of csrc/mlp_inst.hip into a module of its own (hipcc cross-compiles gfx950 without a GPU; cached under
    qinco_amd/_instances/), qinco_load_instance registers it, and qinco_shape_supported then says yes.  (GPU parity of such
    models: tests/test_hip_parity.py::test_arbitrary_geometry_*.)"""
    from qinco_amd.build import ensure_instance, instance_plan
    out3 = (C.c_int32 * 3)()
    assert lib.qinco_padded_shape(100, 256, 500, out3) == 0 and list(out3) == [128, 256, 512]
    assert lib.qinco_padded_shape(100, 128, 256, out3) == 0 and list(out3) == [128, 160, 256]     # projections must survive
    assert lib.qinco_padded_shape(100, 100, 256, out3) == 0 and list(out3) == [128, 128, 256]     # QINCo1: De == D stays
    assert lib.qinco_shape_supported(100, 100, 256) == 1                                          # = the compiled-in C1 shape
    assert instance_plan(128, 256, 512) == (48, 124) and instance_plan(128, 128, 224) == (48, 4476)
    assert instance_plan(224, 224, 320) == (48, 124) and instance_plan(128, 512, 384) == (48, 1236)   # De > 384: 16-row tile form
    with pytest.raises(NotImplementedError):
        instance_plan(128, 1024, 256)
    assert lib.qinco_load_instance(b"/nonexistent/module.so") == -1
    so = ensure_instance(100, 256, 500)
    if so is None:
        assert lib.qinco_shape_supported(100, 256, 500) == 1      # already loaded earlier in this process
    else:
        assert so.exists() and lib.qinco_shape_supported(100, 256, 500) == 1
        mod = C.CDLL(str(so))
        # the capacity argument bounds what the module writes: a short array stays short (round 4: a 4-slot array under a
        # 5-launcher module was 8 bytes of heap corruption and took the whole CPU tier down with it)
        v, fns = (C.c_int32 * 6)(), (C.c_void_p * 8)()
        C.memset(fns, 0xEE, C.sizeof(fns))
        abi = mod.qinco_instance_info(v, fns, 4)
        assert abi > (1 << 16) and list(v)[:5] == [128, 256, 512, 48, 124] and all(fns[i] for i in range(4))
        assert all(fns[i] == 0xEEEEEEEEEEEEEEEE for i in range(4, 8)), "qinco_instance_info wrote past its capacity"
        assert mod.qinco_instance_info(v, fns, 8) == abi and fns[5] == 0xEEEEEEEEEEEEEEEE    # kInstanceNFns = 5 launchers, no more
        assert ensure_instance(100, 256, 500) is None             # second call: nothing to do


def _build_c_host(tmp_path):
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    exe = tmp_path / "c_host"
    r = subprocess.run([gcc, "-O2", "-Wall", "-Werror", "-std=c11", "-D_GNU_SOURCE", "-I", str(ROOT / "include"), str(ROOT / "examples" / "c_host.c"),
                        "-o", str(exe), "-L", str(ROOT / "qinco_amd"), "-lqinco_hip", f"-Wl,-rpath,{ROOT / 'qinco_amd'}", "-lm", "-ldl"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_header_is_plain_c_and_a_c_host_links(tmp_path):
    """include/qinco_hip.h compiles as C11 with -Wall -Werror and examples/c_host.c -- a host with no Python in the process --
    links against libqinco_hip.so.  Without a GPU it must stop at qinco_create with the library's own message."""
    import subprocess
    import torch
    exe = _build_c_host(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("GPU present: test_c_host_runs_end_to_end covers the run")
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "qinco_create" in r.stderr and "no HIP device" in r.stderr, (r.returncode, r.stderr)


@pytest.mark.gpu
def test_c_host_runs_end_to_end(tmp_path):
    """The C host: create from raw arrays, qinco_encode_host -> qinco_gather_codes (one rank) -> qinco_decode_host; the
    beam search must beat random code rows by a wide margin and encode's tracked reconstruction must match decode."""
    import subprocess
    exe = _build_c_host(tmp_path)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    print(r.stdout)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "gather ok" in r.stdout and "model=128x128x256" in r.stdout
