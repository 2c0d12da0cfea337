"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports every symbol that
include/vq3_hip.h declares, and the ctypes binding covers exactly that set (no compute calls without a GPU)."""
import ctypes
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def _declared():
    text = (ROOT / "include" / "vq3_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vq3_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported():
    import __graft_entry__ as g
    g.build()
    from vggt_qwen3_amd import _lib
    lib = ctypes.CDLL(str(_lib.LIB_PATH))
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/vq3_hip.h but not exported"


def test_binding_matches_header():
    from vggt_qwen3_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared()
    lib = _lib.load()
    assert lib.vq3_abi_version() == 1
    assert lib.vq3_target_arch() == b"gfx950"


def test_bad_args_fail_loudly_without_gpu():
    """Argument validation runs on the host before any launch."""
    from vggt_qwen3_amd import _lib
    lib = _lib.load()
    d = _lib.GemmDesc()
    assert lib.vq3_gemm_bf16_nt(ctypes.byref(d), None) != 0
    assert b"null" in lib.vq3_last_error()
