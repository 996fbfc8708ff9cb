"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
that include/idiaptts_amd.h declares (no compute calls without a GPU)."""
import ctypes
import os
import re

from idiaptts_amd import lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "idiaptts_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(itts_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    cdll = ctypes.CDLL(lib.LIB_PATH)
    syms = _header_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(cdll, s), "missing export " + s


def test_python_binding_matches_header():
    assert lib.declared_symbols() == _header_symbols()


def test_scalar_helpers_match_reference_library_values():
    L = lib.load()
    assert L.itts_abi_version() == 1
    # pyworld.get_cheaptrick_fft_size (AudioProcessing.py:60): 1024 @16/22.05/24 kHz, 2048 @44.1/48
    for fs, n in [(16000, 1024), (22050, 1024), (24000, 1024), (44100, 2048), (48000, 2048)]:
        assert L.itts_cheaptrick_fft_size(fs, 71.0) == n
    # pyworld.get_num_aperiodicities (AudioProcessing.py:71)
    for fs, n in [(16000, 1), (22050, 2), (24000, 3), (44100, 5), (48000, 5)]:
        assert L.itts_num_aperiodicities(fs) == n
    # pysptk.util.mcepalpha (AudioProcessing.py:40), values recomputed in SURVEY.md section 2a
    for fs, a in [(16000, 0.41), (22050, 0.455), (24000, 0.466), (44100, 0.544), (48000, 0.554)]:
        assert abs(L.itts_mcep_alpha(fs) - a) < 1e-9
    # LJ001-0001 fixture: 154464 samples @16 kHz -> 1931 frames
    assert L.itts_world_num_frames(154464, 16000, 5.0) == 1931
    assert L.itts_world_synth_length(1931, 16000, 5.0) == 154480


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", str(tmp_path / "nope.so"))
    import pytest
    with pytest.raises(lib.IttsError):
        lib.load()


def test_allreduce_flat_rejects_bad_arguments_without_touching_a_gpu():
    """itts_allreduce_flat (reference hook ModularModelHandlerPyTorch.py:732-735):
    argument errors come back as ITTS_E_INVALID with a message, before RCCL is looked up."""
    L = lib.load()
    assert L.itts_allreduce_flat(None, 16, 0, 0, None, None) == -1
    assert b"itts_allreduce_flat" in L.itts_last_error()
    assert L.itts_allreduce_flat(None, 0, 7, 0, ctypes.c_void_p(1), None) == -1      # unknown dtype
    assert L.itts_allreduce_flat(None, 0, 0, 9, ctypes.c_void_p(1), None) == -1      # unknown reduction
    assert L.itts_allreduce_flat(None, 0, 0, 0, ctypes.c_void_p(1), None) == 0       # empty buffer: no-op
    assert L.itts_comm_init_rank(None, 1, 0, None) == -1
    assert L.itts_comm_destroy(None) == 0
