"""CPU tests of the drop-in boundary: the C-ABI library loads without a GPU and exports
exactly the entry points include/sept.h declares; argument errors come back as status
codes with text, never as crashes.  No compute is launched here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
HEADER = os.path.join(ROOT, "include", "sept.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sept_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported():
    from sept_amd import _lib
    names = _declared()
    assert len(names) >= 8
    for n in names:
        assert hasattr(_lib.lib, n), f"{n} declared in sept.h but not exported"
    # and every symbol the Python host binds is declared in the header
    for n in _lib.SIGNATURES:
        assert n in names, f"{n} bound by the host but missing from sept.h"


def test_abi_version_and_error_text():
    from sept_amd import _lib
    assert _lib.lib.sept_abi_version() >= 1
    h = ctypes.c_void_p()
    st = _lib.lib.sept_mel_plan_create(777, 160, 80, None, None, ctypes.byref(h))
    assert st == -1 and b"null" in _lib.lib.sept_last_error()
    buf = (ctypes.c_float * 4)()
    st = _lib.lib.sept_mel_plan_create(777, 160, 80, buf, buf, ctypes.byref(h))
    assert st == -2 and b"n_fft=777" in _lib.lib.sept_last_error()
    with pytest.raises(_lib.SeptError):
        _lib.check(st, "sept_mel_plan_create")


def test_no_cpu_fallback():
    """CPU tensors must be refused loudly (no silent eager path)."""
    import torch
    from sept_amd import _lib
    with pytest.raises(_lib.SeptError):
        _lib.require_cuda(torch.zeros(3))
    if not torch.cuda.is_available():
        from feature_extraction.audio_feature_extraction import mel_spectrogram
        with pytest.raises(RuntimeError):
            mel_spectrogram(torch.zeros(1, 16000), n_fft=800, feature_len=80)


def test_conv_tile_shapes_at_the_training_widths():
    """The dispatcher's choice of conv tile shape is host-only logic over a table whose ORDER matters: pin the measured
    choices at the widths the two input sizes produce -- 80 mels: 40 / 20 columns, 128 mels: 64 / 32 --
    so that a table edit cannot silently move a training shape to a slower kernel.  (pb, wp, wn, taps per barrier, slices)"""
    from sept_amd import _lib
    want = {
        (32, 64, 40): (2, 8, 1, 0, 1), (64, 128, 20): (2, 4, 2, 0, 2), (64, 32, 40): (2, 8, 1, -2, 2), (128, 64, 20): (2, 8, 1, 0, 4),
        (32, 64, 64): (2, 4, 2, 0, 1), (64, 128, 32): (2, 4, 2, 0, 2), (64, 32, 64): (2, 8, 1, 0, 2), (128, 64, 32): (2, 8, 1, 0, 4),
        (128, 128, 20): (2, 4, 2, 0, 2),
    }
    for (cin, cout, W), shape in want.items():
        for stats in (0, 1):
            out = (ctypes.c_int * 6)()
            assert _lib.lib.sept_conv5x5_variant(W, cin, cout, stats, out) == 0, (cin, cout, W, stats)
            assert tuple(out[:5]) == shape, (cin, cout, W, stats, tuple(out[:5]))
            assert cin == cout or 2 * out[5] <= 160 * 1024            # two workgroups per CU (not the deep model's 128 -> 128)
    out = (ctypes.c_int * 6)()
    assert _lib.lib.sept_conv5x5_variant(20, 48, 64, 0, out) < 0 and b"no kernel" in _lib.lib.sept_last_error()
    # the loader forms exist where the hand-scheduled step uses them
    assert _lib.lib.sept_conv5x5_bnapply_parts(224, 100, 40, 64, 32, 1) > 0 and _lib.lib.sept_conv5x5_bnapply_parts(224, 50, 20, 128, 64, 1) > 0
    assert _lib.lib.sept_conv5x5_act_parts(224, 100, 40, 32, 64, 1) > 0 and _lib.lib.sept_conv5x5_act_parts(224, 100, 40, 64, 128, 1) == 0
    assert _lib.lib.sept_conv5x5_bnapply_parts(224, 99, 40, 64, 32, 1) == 0          # odd height: no whole 2x2 windows


def test_round3_entry_points_refuse_bad_arguments_without_a_gpu():
    """Argument errors of the entry points added in round 3 come back as negative status + text before anything is launched
    (no GPU here): null pointers, forward / data-gradient shapes mixed up, inconsistent epilogue arguments, odd heights."""
    from sept_amd import _lib
    L = _lib.lib
    one = ctypes.c_void_p(16)      # a non-null pointer value that is never dereferenced on these paths
    fz = ctypes.cast(one, ctypes.POINTER(ctypes.c_float)) if False else one
    # sept_conv5x5_dgrad_bnapply: null argument
    st = L.sept_conv5x5_dgrad_bnapply(None, one, one, one, one, one, one, None, one, one, None, None, None, None, None, None, None,
                                      2, 100, 40, 64, 32, None)
    assert st < 0 and b"sept_conv5x5_dgrad_bnapply" in L.sept_last_error()
    # ... a forward shape (cin <= cout) is not a data-gradient launch
    st = L.sept_conv5x5_dgrad_bnapply(one, one, one, one, one, one, one, None, one, one, None, None, None, None, None, None, None,
                                      2, 100, 40, 32, 64, None)
    assert st < 0 and b"data-gradient" in L.sept_last_error()
    # ... an epilogue tensor without its partials buffer
    st = L.sept_conv5x5_dgrad_bnapply(one, one, one, one, one, one, one, None, one, one, one, None, None, one, one, None, None,
                                      2, 100, 40, 64, 32, None)
    assert st < 0 and b"epilogue" in L.sept_last_error()
    # ... an odd height has no whole 2x2 windows
    st = L.sept_conv5x5_dgrad_bnapply(one, one, one, one, one, one, one, None, one, one, None, None, None, None, None, None, None,
                                      2, 99, 40, 64, 32, None)
    assert st < 0 and b"2x2" in L.sept_last_error()
    # sept_conv5x5_forward_act: a data-gradient shape is refused; so is a null BatchNorm
    st = L.sept_conv5x5_forward_act(one, one, one, one, one, None, one, None, one, None, 2, 100, 40, 64, 32, None)
    assert st < 0 and b"forward" in L.sept_last_error()
    st = L.sept_conv5x5_forward_act(one, None, one, one, one, None, one, None, one, None, 2, 100, 40, 32, 64, None)
    assert st < 0 and b"null" in L.sept_last_error()
    # the activation form exists for the conv behind block 1 only
    st = L.sept_conv5x5_forward_act(one, one, one, one, one, None, one, None, one, None, 2, 50, 20, 64, 128, None)
    assert st < 0 and b"no kernel form with this loader" in L.sept_last_error()
    # size queries of shapes without a form answer 0 instead of failing
    assert L.sept_conv5x5_act_parts(0, 100, 40, 32, 64, 1) == 0 and L.sept_conv5x5_bnapply_parts(2, 100, 40, 32, 64, 1) == 0
    out = (ctypes.c_int * 6)()
    assert L.sept_conv5x5_variant(0, 32, 64, 0, out) < 0 and L.sept_conv5x5_variant(40, 32, 64, 0, None) < 0
