"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports every symbol that
include/vq3_hip.h declares, and the ctypes binding covers exactly that set (no compute calls without a GPU)."""
import ctypes
import re

import pytest
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


def test_header_constants_match_the_binding():
    """Sizes the host allocates from a constant of the header."""
    from vggt_qwen3_amd import ops
    text = (ROOT / "include" / "vq3_hip.h").read_text()
    m = re.search(r"#define\s+VQ3_QKPREP_BWD_TOKENS_PER_PART\s+(\d+)", text)
    assert m and int(m.group(1)) == ops.QKPREP_BWD_TOKENS_PER_PART


def test_bad_args_fail_loudly_without_gpu():
    """Argument validation runs on the host before any launch."""
    from vggt_qwen3_amd import _lib
    lib = _lib.load()
    d = _lib.GemmDesc()
    assert lib.vq3_gemm_bf16_nt(ctypes.byref(d), None) != 0
    assert b"null" in lib.vq3_last_error()


def _tile_order(lib, M, N, bm, bn, wg):
    import numpy as np
    mt, nt = (M + bm - 1) // bm, (N + bn - 1) // bn
    o = np.zeros((mt * nt, 2), np.int32)
    xm, nb = ctypes.c_int32(), ctypes.c_int32()
    assert lib.vq3_gemm_tile_order(M, N, bm, bn, wg, ctypes.byref(xm), ctypes.byref(nb), o.ctypes.data_as(ctypes.c_void_p)) == 0
    return o, xm.value, nb.value, mt, nt


def test_gemm_tile_order_is_a_bijection_and_blocks_per_xcd():
    """The workgroup -> tile map of the GEMM kernels (gemm_common.h: tile_coords_id, host twin behind vq3_gemm_tile_order): every tile
    exactly once for any grid (ragged XCD rectangles, ragged last band, grids smaller than the chip), and - the point of the banded
    walk - the 32 tiles an XCD has in flight together on the tall VGGT-tower shapes form a block of few A row panels x few B panels
    (round 2's m-fastest walk had 25-32 different A panels per n-tile in flight)."""
    from vggt_qwen3_amd import _lib
    lib = _lib.load()
    shapes = [(49392, 4096, 256, 256, 1), (49392, 4096, 128, 256, 1), (49392, 3072, 128, 256, 1), (49392, 1024, 256, 128, 1),
              (9600, 19456, 256, 256, 1), (1200, 2560, 128, 128, 2), (2560, 9728, 256, 256, 1), (300, 520, 128, 128, 2),
              (77, 200, 128, 128, 2), (6174, 1024, 128, 128, 2), (152000, 2560, 128, 128, 1), (112, 151937, 128, 128, 2)]
    for M, N, bm, bn, wg in shapes:
        o, xm, nb, mt, nt = _tile_order(lib, M, N, bm, bn, wg)
        assert xm in (1, 2, 4, 8) and nb >= 1
        ids = (o[:, 0].astype("int64") * nt + o[:, 1]).tolist()
        assert sorted(ids) == list(range(mt * nt)), (M, N, bm, bn)
        assert int(o[:, 0].max()) == mt - 1 and int(o[:, 1].max()) == nt - 1
    # tall tower shapes (measured rule, tools/sweep_tile_order.sh): every XCD owns a contiguous strip of M - activation panels cross the
    # fabric once - and walks it in bands of 2 n-tiles, so the 32 tiles it has in flight share <= 17 A panels and 2 B panels (4 where a window straddles two bands)
    for M, N, bm, bn in ((49392, 4096, 256, 256), (49392, 4096, 128, 256), (49392, 3072, 128, 256)):
        o, xm, nb, mt, nt = _tile_order(lib, M, N, bm, bn, 1)
        assert xm == 8 and nb == 2
        strips = []
        for xcd in range(8):
            mine = o[xcd::8]                              # workgroup b runs on XCD b % 8, in order
            strips.append((int(mine[:, 0].min()), int(mine[:, 0].max())))
            for r in range(0, min(len(mine), 320) - 32, 32):
                blk = mine[r:r + 32]
                na, nbp = len(set(blk[:, 0].tolist())), len(set(blk[:, 1].tolist()))
                assert nbp <= 4 and na <= 17, (M, N, bm, bn, xcd, r, na, nbp)      # (a window may straddle two bands)
        # strips of M, one per XCD (an XCD's share of the walk is nwg / 8 tiles, a region is ceil(mt / 8) rows: neighbours overlap by
        # the few rows the two counts differ by, never more than a strip)
        strips.sort()
        width = -(-mt // 8)
        assert all(lo2 > lo1 and hi1 - lo1 < 2 * width for (lo1, hi1), (lo2, _) in zip(strips, strips[1:])), strips


def test_gemm_split_plan_host_side():
    """vq3_gemm_split_plan (cfg 25): which launches cut their last round of 256 x 256 tiles along K, and how - host arithmetic only."""
    from vggt_qwen3_amd import ops
    assert ops.gemm_split_plan(9600, 2560, 4096) == (256, 124, 2)          # o-projection of a pass of 8 micro-batches: 380 tiles
    assert ops.gemm_split_plan(1200, 2560, 2560) == (0, 50, 4)             # one micro-batch: 50 tiles x 4 slices of 10 K tiles
    assert ops.gemm_split_plan(12000, 2560, 4096)[2] == 0                  # 470 tiles: the last round is 84 % full
    assert ops.gemm_split_plan(9600, 2560, 512)[2] == 0                    # 8 K tiles: nothing to split
    assert ops.gemm_split_plan(2048, 2048, 4096)[2] == 4                   # 64 tiles x 4
    for (M, N, K) in [(9600, 2560, 9728), (2000, 2560, 1088), (300, 300, 1024), (40000, 1024, 1024)]:
        full, rem, sl = ops.gemm_split_plan(M, N, K)
        tiles = -(-M // 256) * -(-N // 256)
        if sl:
            per = -(-(K // 64) // sl)
            assert full + rem == tiles and full % 256 == 0 and 2 <= sl <= 4 and rem * sl <= 256
            assert (K // 64) - (sl - 1) * per >= 1 and K // 64 // sl >= 8


def test_persistent_decode_kernel_has_no_private_segment(tmp_path):
    """vq3_qwen_decode_layers is launched once per token: any scratch (a spilled register) makes the runtime set up a private segment for
    every wave slot at each launch - measured 0.3 ms per token, more than the kernel gains. Read the kernel's
    metadata out of the object the library was linked from."""
    import shutil
    import subprocess
    obj = ROOT / "vggt_qwen3_amd" / "csrc" / "build" / "decode_layers.o"
    llvm = Path("/opt/rocm/lib/llvm/bin")
    if not obj.exists() or not (llvm / "clang-offload-bundler").exists():
        pytest.skip("needs the in-tree build directory and the ROCm LLVM tools")
    fat, co = tmp_path / "fat.bin", tmp_path / "dl.co"
    subprocess.run([str(llvm / "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", str(obj)], check=True)
    subprocess.run([str(llvm / "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}",
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
    notes = subprocess.run([str(llvm / "llvm-readelf"), "--notes", str(co)], check=True, capture_output=True, text=True).stdout
    kern = [b for b in notes.split(".name:") if "decode_layers_kernel" in b.split("\n")[0]]
    assert kern, "decode_layers_kernel not found in the code object"
    for b in kern:
        seg = [ln for ln in b.split("\n") if ".private_segment_fixed_size:" in ln]
        assert seg and int(seg[0].split(":")[1]) == 0, seg
