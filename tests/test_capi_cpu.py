"""The C-ABI library loads on a machine without a GPU, exports exactly what
include/auvsim.h declares, and fails loudly (never falls back) when asked to
compute without a device."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from gym_auv_amd import _lib, params

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "auvsim.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(auv_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _declared_functions() == sorted(_lib.EXPORTS)


def test_library_exports_every_declared_symbol(lib):
    for name in _declared_functions():
        assert hasattr(lib, name), f"libauvsim.so does not export {name}"
    assert b"gfx950" in lib.auv_version()


def test_config_struct_layout_matches_header():
    # 8 int32 + 7 double + risk[3] + reward[4] + model[16] + contact_dist
    assert C.sizeof(_lib.AuvConfig) == 8 * 4 + (7 + 3 + 4 + 16 + 1) * 8


def _header_config_fields():
    """(name, C type, array length or 0) of every member of `auv_config`, in order, from the header text."""
    text = open(os.path.join(ROOT, "include", "auvsim.h")).read()
    body = re.search(r"typedef struct auv_config\s*\{(.*?)\}\s*auv_config;", text, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    out = []
    for decl in body.split(";"):
        m = re.match(r"\s*(int32_t|double)\s+(.*)$", decl.strip(), flags=re.S)
        if not m:
            continue
        for item in m.group(2).split(","):
            a = re.match(r"\s*(\w+)\s*(?:\[(\w+)\])?\s*$", item)
            n = a.group(2) or "0"
            if not n.isdigit():      # a length given by a #define of the header
                n = re.search(r"#define\s+%s\s+(\d+)" % n, text).group(1)
            out.append((a.group(1), m.group(1), int(n)))
    return out


def test_binding_and_integration_stub_name_the_headers_config_fields():
    """The struct in the ctypes binding and the one INTEGRATION.md shows a maintainer must both be the header's:
    same member names, types and array lengths, in the same order."""
    want = _header_config_fields()
    assert len(want) == 19 and want[7] == ("n_path_segments", "int32_t", 0)
    ctype = {"int32_t": C.c_int32, "double": C.c_double}
    got = []
    for name, t in _lib.AuvConfig._fields_:
        n = getattr(t, "_length_", 0)
        got.append((name, {C.c_int32: "int32_t", C.c_double: "double"}[t._type_ if n else t], n))
    assert got == want
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    stub = doc[doc.index("class auv_config(ctypes.Structure)"):]
    stub = stub[:stub.index("]\n") + 1]
    fields = re.findall(r'\("(\w+)",\s*ctypes\.(c_int32|c_double)(?:\s*\*\s*(\d+))?\)', stub)
    shown = [(n, {"c_int32": "int32_t", "c_double": "double"}[t], int(k or 0)) for n, t, k in fields]
    assert shown == want, "INTEGRATION.md's auv_config stub differs from include/auvsim.h"
    assert ctype  # (both maps above cover every type the header uses)


def test_buffer_enum_matches_header():
    text = open(os.path.join(ROOT, "include", "auvsim.h")).read()
    enum = dict(re.findall(r"(AUV_BUF_[A-Z_]+)\s*=\s*(\d+)", text))
    for name, idx in _lib.BUFFER_NAMES.items():
        assert int(enum["AUV_BUF_" + name.upper()]) == idx
    assert int(enum["AUV_BUF_COUNT"]) == _lib.BUF_COUNT


def test_create_rejects_bad_arguments_before_touching_a_device(lib):
    h = C.c_void_p()
    cfg = _lib.AuvConfig()
    cfg.abi_version = 999
    assert lib.auv_create(C.byref(cfg), 0, C.byref(h)) == -1
    assert b"abi_version" in lib.auv_last_error(None)
    cfg.abi_version = _lib.ABI_VERSION
    cfg.n_envs = 0
    cfg.n_rays = 180
    assert lib.auv_create(C.byref(cfg), 0, C.byref(h)) == -1
    assert not h.value


def test_no_gpu_means_error_not_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the no-device path cannot be exercised")
    from gym_auv_amd.sim import AuvSim

    with pytest.raises(_lib.AuvError) as ei:
        AuvSim(params.make_config(4, 1, 0))
    assert ei.value.code == -2


def test_dlpack_record_layout():
    # the mirror of the C record the consumer reads: 48-byte DLTensor, then ctx + deleter
    from gym_auv_amd import dlpack

    assert C.sizeof(dlpack.DLTensor) == 48
    assert C.sizeof(dlpack.DLManagedTensor) == 64
    assert dlpack.DLTensor.shape.offset == 24 and dlpack.DLTensor.byte_offset.offset == 40


def test_model_constants_invert_the_mass_matrix():
    mp = params.model_constants()
    M = np.array([[25.8, 0, 0], [0, 33.8, mp[6]], [0, mp[6], 2.76]])     # m - X_udot, m - Y_vdot, m x_g, I_z - N_rdot
    assert mp[5] == 23.8 and mp[6] == 23.8 * 0.046 and mp[7] == 0.0
    Minv = np.array([[mp[0], 0, 0], [0, mp[1], mp[2]], [0, mp[3], mp[4]]])
    assert np.allclose(M @ Minv, np.eye(3), atol=1e-14)
