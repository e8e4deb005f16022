"""CPU-only: the C-ABI library loads and exports every symbol include/imdbn_engine.h declares
(no compute calls -- there is no GPU here), and size helpers are consistent."""
import os
import re

from imdbn.engine import native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "imdbn_engine.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(imdbn_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = native.lib()
    syms = _declared_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in the header but not exported"
        assert s in native.SIGNATURES, f"{s} has no ctypes signature in imdbn/engine/native.py"
    assert sorted(native.SIGNATURES) == syms, "ctypes binding lists symbols the header does not declare"


def test_version_and_sizes():
    lib = native.lib()
    assert lib.imdbn_version() == native.ABI_VERSION == 4      # include/imdbn_engine.h IMDBN_ABI_VERSION
    assert lib.imdbn_ws_bytes(10000, 1500, 64) > 0
    assert lib.imdbn_ws_bytes(10000, 1500, 64) <= 128 << 20        # scratch stays small next to 288 GB
    assert lib.imdbn_ws_bytes(0, 10, 1) == 0
    assert lib.imdbn_ws_bytes(784, 256, 32) == lib.imdbn_ws_bytes(784, 256, 64)   # batch is padded to 64


def test_struct_sizes_match_header_layout():
    import ctypes as C
    assert C.sizeof(native.RbmDesc) == 8 * 7 + 4 * 4 + 4 * 8
    assert C.sizeof(native.Rng) == 8 + 8 * 3 + 8 * 4 + 8 * 3 + 8
    assert C.sizeof(native.ChainStep) == 24
    assert C.sizeof(native.CdOpts) == 88            # 9 x 4 B + pad + next_data, ld_next, next_slot, data_slot, data_binary, next_binary, fwd_out, ld_fwd


def test_options_handle_is_separate_from_the_process_defaults():
    """imdbn_options: a per-caller copy of the knobs, bound to the calling thread with imdbn_use_options."""
    import ctypes as C
    lib = native.lib()
    h = C.c_void_p(lib.imdbn_options_create())
    assert h
    assert lib.imdbn_options_set(h, b"no_k1s", 1) == 0 and lib.imdbn_options_set(h, b"k2s_rows", 24) == 0
    assert lib.imdbn_options_set(h, b"k2s_rows", 23) != 0            # same validation as imdbn_set_option
    assert lib.imdbn_options_set(h, b"no_such_knob", 1) != 0
    assert lib.imdbn_options_set(h, b"dbg", 1) != 0                  # the debug stamps are process-wide
    assert lib.imdbn_options_set(None, b"no_k1s", 1) != 0
    assert lib.imdbn_use_options(h) == 0 and lib.imdbn_use_options(None) == 0
    assert lib.imdbn_use_options(h) == 0
    lib.imdbn_options_destroy(h)                                     # destroying the bound handle unbinds it
    assert lib.imdbn_set_option(b"no_k1s", 0) == 0


def test_no_gpu_call_reports_nodevice_as_exception():
    import torch
    import pytest
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from imdbn.engine.hip_engine import HipEngine
    eng = HipEngine()
    with pytest.raises(native.EngineError):
        eng.device_info()
