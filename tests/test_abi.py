"""CPU suite: the C-ABI library loads and exports every symbol include/gtsc.h
declares; host-only helpers agree with the oracle.  No compute calls here."""
import ctypes as C
import os
import re
import subprocess

import pytest

import pygtsc as G
import pyoracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "gtsc.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(gtsc_[a-z_0-9]+)\s*\(", hdr)))


def test_header_and_binding_list_the_same_symbols():
    assert declared_symbols() == sorted(G.SYMBOLS)


def test_library_exports_every_declared_symbol():
    lib = G.lib()
    for name in declared_symbols():
        assert hasattr(lib, name), name
    out = subprocess.check_output(["nm", "-D", "--defined-only", G.LIB_PATH], text=True)
    exported = set(re.findall(r"\bT (gtsc_[a-z_0-9]+)", out))
    assert exported == set(declared_symbols())


def test_host_only_helpers():
    lib = G.lib()
    assert lib.gtsc_version() >= 1000
    for p in (0, 1, 2, 63, 64, 120, 7200, 100000):
        assert G.max_words(p) == O.max_words(p)
    assert G.max_words(120) == 216


def test_argument_errors_without_touching_a_gpu():
    lib = G.lib()
    assert lib.gtsc_sync(None) == G.ERR_ARG
    assert b"NULL" in lib.gtsc_last_error()
    assert lib.gtsc_create(0, None) == G.ERR_ARG
    assert lib.gtsc_encode_device(None, 0, 0, None, None, None, None, 0, None, None, 0, None, None) == G.ERR_ARG
    assert lib.gtsc_destroy(None) == G.OK


def test_create_fails_cleanly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    rc = G.lib().gtsc_create(0, C.byref(h))
    assert rc == G.ERR_HIP and not h.value
    with pytest.raises(G.GtscError):
        G.Context(0)
