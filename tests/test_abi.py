"""The C-ABI shared library: loads, exports every symbol include/nwe.h declares, and its host-only paths
(context, validation, packing) behave; no device work here."""
import ctypes as C
import os
import re
import sys

import numpy as np
import pytest

import nwe_amd
from nwe_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "nwe.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nwe_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = declared_symbols()
    assert "nwe_render" in names and "nwe_set_network" in names and len(names) >= 14
    lib = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/nwe.h but not exported"
    assert sorted(_lib.SYMBOLS) == names        # the ctypes binding covers exactly the header


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libnwe_hip.so"))
    with pytest.raises(RuntimeError, match="is missing"):
        nwe_amd.Renderer(host_only=True)


def test_host_only_context_validation():
    r = nwe_amd.Renderer(host_only=True)
    sd = nwe_amd.synthetic.make_state_dict(1, 4, 128)
    assert r.set_network(0, sd) == (4, 128, 63, 27, -1)
    # checkpoint spelling without the leading underscore (handler.py:150-164) is accepted
    assert r.set_network(1, {k[1:]: v for k, v in sd.items()}) == (4, 128, 63, 27, -1)
    bad = dict(sd)
    bad["_rgb_linear.weight"] = np.zeros((3, 65), np.float32)
    with pytest.raises(ValueError, match="_rgb_linear"):
        r.set_network(0, bad)
    with pytest.raises(NotImplementedError):
        r.set_network(0, nwe_amd.synthetic.make_state_dict(1, 4, 512))      # width > 256
    r.set_sampling(64, 128)
    with pytest.raises(NotImplementedError):
        r.set_sampling(300, 0)
    # a host-only context refuses to render instead of falling back to anything
    lib = _lib.load()
    out = _lib.Outputs()
    rc = lib.nwe_render_rays(r._ctx, None, 0, 0, C.byref(out), None)
    assert rc == _lib.NWE_ERR_STATE and b"host-only" in lib.nwe_last_error(r._ctx)
    r.close()


def test_flops_per_eval_match_baseline():
    r = nwe_amd.Renderer(host_only=True)
    r.set_network(0, nwe_amd.synthetic.make_state_dict(1, 8, 256))
    r.set_network(1, nwe_amd.synthetic.make_state_dict(1, 4, 128))
    assert r.flops_per_eval(0) == 1186816 and r.flops_per_eval(1) == 167680


def test_stream_layout_properties():
    """Weight stream invariants the kernel relies on: tile-aligned, (hi, lo) tile pairs, (hi + lo) / scale == weight,
    a separate bias table with one 32-float row per chunk."""
    sd = nwe_amd.synthetic.make_state_dict(9, 4, 128)
    r = nwe_amd.Renderer(host_only=True)
    r.set_network(0, sd)
    s, bias, scale = r.packed_stream(0), r.packed_bias(0), r.packed_scale(0)
    assert s.size % 1024 == 0 and np.log2(scale) == int(np.log2(scale))
    assert np.array_equal(bias[0], sd["_pts_linears.0.bias"][:32])
    hi = s[0:1024].view(np.float16).astype(np.float64)
    lo = s[1024:2048].view(np.float16).astype(np.float64)
    # lane 0 (row 0, half 0), element 0 of k-step 0 is gamma slot 0 = sin(2^0 x) = column 3 of the encoding
    w = sd["_pts_linears.0.weight"]
    assert abs((hi[0] + lo[0]) / scale - w[0, 3]) < 1e-7 * max(1, abs(w[0, 3]))
    assert abs((hi[1] + lo[1]) / scale - w[0, 6]) < 1e-7                      # slot 1 = cos(2^0 x) = column 6
    assert np.abs(hi).max() < 2.0 ** 15                                        # scaled into fp16 range with headroom
    nz = lo[lo != 0]
    assert (np.abs(nz) >= 2.0 ** -14).mean() > 0.99                            # lo halves are fp16-normal
    # folded stream: behind the chunks' bias rows come the dot rows of _alpha_linear (W/32 rows of weights in the row order of
    # the last trunk layer's tiles, then its bias); the last chunk is the rgb head tile, whose rows 4..7 repeat rows 0..3 so
    # that both lane halves see the outputs
    n_dot = 128 // 32 + 1
    assert np.array_equal(bias[-n_dot:-1].reshape(-1), sd["_alpha_linear.weight"][0]) and bias[-1][0] == sd["_alpha_linear.bias"][0]
    assert not bias[-1][1:].any()
    rgb_row = bias[-n_dot - 1]
    assert np.array_equal(rgb_row[:3], sd["_rgb_linear.bias"]) and np.array_equal(rgb_row[4:7], sd["_rgb_linear.bias"])


def test_kernel_owns_m0():
    """nwe_mfma_kernels.h keeps the LDS-DMA destination in M0 across statements (one write per group of pieces), which is
    sound only while hipcc emits no M0 use of its own in that kernel: tools/check_m0.py disassembles the SHIPPED library and
    checks that every instruction touching m0 in the render kernels is one of ours.  A missing disassembler fails the
    test (the same check runs in __graft_entry__.build(), so a library that breaks the invariant does not get built)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_m0
    assert check_m0.check(_lib.LIB_PATH) > 1000


def test_outputs_struct_guard_and_integration_stub_layout():
    """nwe_outputs carries its own size: a caller built against another version of include/nwe.h is refused instead of
    having its pointers misread; and the struct printed in INTEGRATION.md has exactly the fields of the header."""
    import ctypes as C
    import re
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "nwe.h")).read()
    body = header[header.index("typedef struct nwe_outputs {"):header.index("} nwe_outputs;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"(?:float|uint32_t|uint64_t)\s*\*?\s*(\w+)\s*;", body)
    assert fields == ["struct_bytes"] + list(_lib.OUTPUT_FIELDS), fields
    stub = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    stub = stub[stub.index("class Outputs(C.Structure)"):stub.index("lib.nwe_last_error.restype")]
    assert re.findall(r'"(\w+)"', stub) == fields
    ctx = C.c_void_p()
    assert lib.nwe_create(C.byref(ctx), -1) == 0
    o = _lib.Outputs()
    assert o.struct_bytes == C.sizeof(_lib.Outputs) == 8 * (1 + len(_lib.OUTPUT_FIELDS))
    o.struct_bytes -= 8                                          # "an older header"
    assert lib.nwe_render_rays(ctx, None, 0, 0, C.byref(o), None) == _lib.NWE_ERR_INVALID
    assert b"struct_bytes" in lib.nwe_last_error(ctx)
    lib.nwe_destroy(ctx)


def test_header_is_usable_from_plain_c(tmp_path):
    """include/nwe.h compiled as C99 by gcc, linked against the library, exercised on a host-only context
    (tests/c/abi_smoke.c): the boundary is a C ABI, not a C++ one."""
    import subprocess
    exe = tmp_path / "abi_smoke"
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "abi_smoke.c"),
                    "-o", str(exe), "-L", libdir, "-lnwe_hip", "-lm", f"-Wl,-rpath,{libdir}"], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and "abi_smoke ok" in out.stdout, out.stderr


def test_host_code_under_sanitizers():
    """`make asan`: the HOST half of nwe_abi.hip (validation, the fp32 and MFMA packers with the fp64 fold, the copies out)
    built with AddressSanitizer + UndefinedBehaviorSanitizer and driven by tests/c/abi_smoke.c on a host-only context.  GPU
    sanitizers do not exist on this pool; the device code is covered by the parity tests."""
    import shutil
    import subprocess
    if not shutil.which("hipcc") and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc in this environment")
    csrc = os.path.join(ROOT, "nerf-workspaces-explorer_amd", "csrc")
    out = subprocess.run(["make", "-C", csrc, "asan"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "abi_smoke ok" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])
    assert "AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr


def test_graft_entry_build_runs():
    """The driver's "does it build" step, end to end: `make` (a no-op when the library is current), the M0 check, the oracle
    import and the packed-stream sanity check of __graft_entry__.build() - a stale constant there fails HERE, not at round end."""
    import subprocess
    out = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build()"], cwd=ROOT, capture_output=True, text=True, timeout=1800)
    assert out.returncode == 0 and "build ok" in out.stdout, (out.stdout[-1500:], out.stderr[-1500:])
