"""CPU-side checks of the drop-in boundary: the shared library loads and exports
every symbol include/miso_hip.h declares (no compute calls without a GPU)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "miso_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(miso_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from miso_amd import _lib
    names = _declared()
    assert len(names) >= 12
    lib = _lib.load()
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"{n} declared in miso_hip.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature in miso_amd/_lib.py"
    assert set(_lib.SIGNATURES) == set(names)
    assert b"gfx950" in lib.miso_version()
    # the library was built from THIS tree: miso_version() embeds a hash of the kernel sources
    import importlib.util
    spec = importlib.util.spec_from_file_location("srchash", os.path.join(ROOT, "miso_amd", "csrc", "srchash.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert lib.miso_version().decode().endswith("src=" + mod.source_hash()), \
        "libmiso_hip.so is stale: rebuild with `make -C miso_amd/csrc`"
    assert lib.miso_error_string(2001) == b"bad argument"


def test_struct_layout_matches_header():
    from miso_amd import _lib
    # sizes implied by the C declarations (LP64)
    assert ctypes.sizeof(_lib.Level) == 8 + 8 + 4 * 4 + 8 * 4 + 8      # ... + grad_touched
    assert ctypes.sizeof(_lib.Grid) == 4 + 4 + 12 + 12 + 4 + 4 + 8 * ctypes.sizeof(_lib.Level)
    assert ctypes.sizeof(_lib.Mlp) == 16 + 8 * 4 + 8 * 4


def test_ops_refuse_cpu_tensors():
    """No silent CPU fallback: the product ops raise on CPU tensors."""
    from miso_amd import ops
    f = torch.zeros(1, 4, 3, 3, 3)
    x = torch.zeros(5, 3)
    meta = ops.GridMeta.from_bound([[-1, 1]] * 3)
    with pytest.raises(RuntimeError, match="HIP device only"):
        ops.encode(x, [f], meta)
    with pytest.raises(RuntimeError, match="HIP device only"):
        ops.adam_dense_(f, f.clone(), f.clone(), f.clone(), 1, 1e-3)


def test_argument_validation_without_gpu():
    """Entry points validate arguments before touching the device."""
    from miso_amd import _lib
    lib = _lib.load()
    g = _lib.Grid()
    g.n_levels = 0
    assert lib.miso_encode_fwd(ctypes.byref(g), None, 0, None, 0, None) == 2001
    g.n_levels = 1
    g.level[0].C = 4; g.level[0].X = 2; g.level[0].Y = 2; g.level[0].Z = 2
    g.level[0].sC = 1; g.level[0].sX = 4; g.level[0].sY = 8; g.level[0].sZ = 16
    assert lib.miso_encode_fwd(ctypes.byref(g), None, 0, None, 4, None) == 2001  # data NULL
    g.flags = 128
    assert lib.miso_encode_fwd(ctypes.byref(g), None, 0, None, 4, None) == 2001  # bad flag
    m = _lib.Mlp()
    m.in_dim, m.hidden_dim, m.out_dim, m.n_linear = 24, 64, 1, 3
    assert lib.miso_mlp_packed_floats(ctypes.byref(m)) > 0
    assert lib.miso_sdf_mask_words(ctypes.byref(m)) == 4
    m.hidden_dim = 48
    assert lib.miso_mlp_packed_floats(ctypes.byref(m)) == 0


def test_round2_entry_points_validate_arguments_without_gpu():
    """The entry points added for the captured trainer step, the tracker and the batch prologue refuse bad arguments
    before any launch; miso_adam_scalars_table is a host function and can be checked against the formula here."""
    import math
    from miso_amd import _lib
    lib = _lib.load()
    E = 2001
    # step scalars: {1 - b1, b2, 1 - b2, -lr / (1 - b1^t), sqrt(1 - b2^t), eps} as torch.optim.Adam computes them
    host = torch.empty((4, 6), dtype=torch.float32)
    assert lib.miso_adam_scalars_table(1e-3, 0.9, 0.999, 1e-8, 1, 4, ctypes.c_void_p(host.data_ptr())) == 0
    for t in range(1, 5):
        want = [1 - 0.9, 0.999, 1 - 0.999, -(1e-3 / (1 - 0.9 ** t)), math.sqrt(1 - 0.999 ** t), 1e-8]
        assert torch.allclose(host[t - 1], torch.tensor(want, dtype=torch.float32), rtol=1e-6, atol=0)
    assert lib.miso_adam_scalars_table(1e-3, 0.9, 0.999, 1e-8, 0, 4, ctypes.c_void_p(host.data_ptr())) == E
    assert lib.miso_adam_scalars_table(1e-3, 0.9, 0.999, 1e-8, 1, 4, None) == E
    assert lib.miso_adam_bump(None, None, None) == E
    assert lib.miso_adam_step_dev(None, None, None, None, None, None, 8, None, 1, None, 0, None, None) == E
    assert lib.miso_adam_touched(None, None, None, None, None, None, 8, 1e-3, 0.9, 0.999, 1e-8, 1, 0, None, None) == E
    assert lib.miso_adam_touched(None, None, None, None, None, None, 0, 1e-3, 0.9, 0.999, 1e-8, 0, 0, None, None) == E   # step 0
    assert lib.miso_mapping_batch(None, None, 1, None, 1, None, None, None, None, None, None, None, 0, 0, None, None, 0, None) == E
    assert lib.miso_mapping_loss_rows(7, 1.0, 0.0, 0.0, None, None, 0, None, None, None) == E
    assert lib.miso_mapping_loss_rows(1, 1.0, 0.0, 0.0, None, None, 4, None, None, None) == E
    assert lib.miso_lm_track_step(None, None, None, None, None) == E
    assert lib.miso_track_adam_step(None, None, None, None, None) == E
    a = _lib.LmTrack()
    g = _lib.Grid()
    g.n_levels = 1
    assert lib.miso_lm_track_step(ctypes.byref(g), None, None, ctypes.byref(a), None) == E      # no pose pointers
    t = _lib.TrackAdam()
    assert lib.miso_track_adam_step(ctypes.byref(g), None, None, ctypes.byref(t), None) == E
    # struct sizes the header implies (LP64)
    assert ctypes.sizeof(_lib.LmTrack) == 4 * 8 + 3 * 8 + 8 + 8 + 8 + 8 + 4 * 8 + 4 + 4 + 4 + 4 + 9 * 8
    assert ctypes.sizeof(_lib.TrackAdam) == ctypes.sizeof(_lib.LmTrack) + 4 + 4 + 4 + 4 + 8 + 8 + 8 + 8 + 8 + 8


def test_gradient_plans_are_host_logic():
    """Which levels the binned backward pulls, pushes or scatters is decided on the host from the grid shape, the batch
    size and the flags (miso_grad_pull_levels / miso_sdf_bwd_push_levels): no device call, so it is checked here."""
    from miso_amd import _lib
    lib = _lib.load()

    def grid(sizes, C, flags=0):
        g = _lib.Grid()
        g.n_levels = len(sizes)
        g.flags = flags
        for a in range(3):
            g.bound_min[a], g.bound_max[a] = -1.0, 1.0
        for l, (x, y, z) in enumerate(sizes):
            lv = g.level[l]
            lv.C, lv.X, lv.Y, lv.Z = C, x, y, z
            lv.sC, lv.sX, lv.sY, lv.sZ = 1, C, C * x, C * x * y          # channels-last
            lv.data = None
            lv.grad = 0x100000 * (l + 1)                                 # never dereferenced by the planners
        return g

    T = 16
    cfg2 = grid([(32, 32, 32), (64, 64, 64), (128, 128, 128)], 8)
    assert lib.miso_grad_pull_levels(ctypes.byref(cfg2), T) == 0b111
    assert lib.miso_sdf_bwd_push_levels(ctypes.byref(cfg2), T, 262144) == 0          # 64 samples a tile: pulled
    scannet = grid([(40, 20, 40), (200, 100, 200)], 4)
    assert lib.miso_grad_pull_levels(ctypes.byref(scannet), T) == 0b01                # 13 x 7 x 13 bricks: scattered
    assert lib.miso_sdf_bwd_push_levels(ctypes.byref(scannet), T, 540000) == 0b01     # a crowd on the coarse level
    assert lib.miso_sdf_bwd_push_levels(ctypes.byref(scannet), T, 54000) == 0
    crowded = grid([(40, 20, 40), (200, 100, 200)], 4, flags=_lib.F_CROWDED)
    assert lib.miso_sdf_bwd_push_levels(ctypes.byref(crowded), T, 54000) == 0b01      # the caller's hint
    assert lib.miso_sdf_bwd_push_levels(ctypes.byref(crowded), T, 1000) == 0
    ncd = grid([(120, 120, 20), (600, 600, 100)], 4)
    assert lib.miso_grad_pull_levels(ctypes.byref(ncd), T) == 0b01                   # 38 vertices per tile and axis
    border = grid([(32, 32, 32)], 8, flags=2)                                         # padding_mode='border'
    assert lib.miso_grad_pull_levels(ctypes.byref(border), T) == 0


def test_struct_sizes_and_offsets_equal_the_headers_as_a_c_compiler_sees_it(tmp_path):
    """include/miso_hip.h compiled by gcc: sizeof of every struct the ctypes binding mirrors and the offsets of the
    fields added last (a binding that drifts from the header corrupts arguments silently)."""
    import ctypes
    import os
    import subprocess
    from miso_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "layout.c"
    names = [("miso_level_t", _lib.Level), ("miso_grid_t", _lib.Grid), ("miso_mlp_t", _lib.Mlp),
             ("miso_sorted_t", _lib.Sorted), ("miso_align_pair_t", _lib.AlignPair), ("miso_align_t", _lib.Align),
             ("miso_lm_track_t", _lib.LmTrack), ("miso_track_adam_t", _lib.TrackAdam),
             ("miso_adam_tensor_t", _lib.AdamTensor)]
    offs = [("miso_align_t", "poses_ready", _lib.Align), ("miso_align_t", "state", _lib.Align),
            ("miso_sorted_t", "pull_queue_ints", _lib.Sorted), ("miso_level_t", "grad_touched", _lib.Level)]
    body = "".join(f'  printf("%zu\\n", sizeof({n}));\n' for n, _ in names)
    body += "".join(f'  printf("%zu\\n", offsetof({n}, {f}));\n' for n, f, _ in offs)
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "miso_hip.h"\nint main(void) {\n' + body + "  return 0;\n}\n")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(v) for v in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    want = [ctypes.sizeof(c) for _, c in names] + [getattr(c, f).offset for _, f, c in offs]
    assert got == want, list(zip([n for n, _ in names] + [f"{n}.{f}" for n, f, _ in offs], got, want))
    assert _lib is not None


def test_tile_codes_pack_and_choose():
    """MISO_TILES_XYZ on the Python side (ops.pack_tiles / n_tiles / choose_tiles): a count stays a count, per-axis counts
    are packed as the header's macro packs them, and the policy follows the finest level within what the matrix-core
    pull accepts (3 size >= 2 tiles for every level)."""
    import torch
    from miso_amd import ops
    assert ops.pack_tiles(16) == 16 and ops.pack_tiles((16, 16, 16)) == 16 and ops.n_tiles(16) == 4096
    code = ops.pack_tiles((25, 13, 25))
    assert code == 25 | (13 << 8) | (25 << 16) and ops.n_tiles(code) == 25 * 13 * 25
    assert ops.pack_tiles((32, 32, 32)) == 32 | (32 << 8) | (32 << 16)          # cubic but beyond 16: packed
    f = lambda z, y, x: torch.empty(1, 4, z, y, x)                                # noqa: E731
    assert ops.choose_tiles([f(32, 32, 32), f(128, 128, 128)]) == 16             # cfg-2: 8 vertices per tile already
    assert ops.choose_tiles([f(40, 20, 40), f(200, 100, 200)]) == ops.pack_tiles((25, 16, 25))      # ScanNet submap
    # a coarse level of 16 vertices caps the binning at 24 tiles: the 200-vertex axis cannot be owned, stays at 16
    assert ops.choose_tiles([f(16, 16, 16), f(200, 100, 200)]) == 16
    assert ops.choose_tiles([f(20, 120, 120), f(100, 600, 600)]) == 16           # Newer College: too fine even for 32


def test_train_kernel_lds_fits_for_every_fused_shape():
    """the scattering form of the one-launch training kernel needs the weights + four d-feat tiles + four
    record blocks in LDS; sdf_train_supported routes a shape that would not fit to the two-launch path.  The formula
    (ops.sdf_train_lds_bytes) follows sdf_fused.hip's PackLayout; every shape the library instantiates fits 160 KB."""
    from miso_amd import ops
    shapes = [(4, 1, 32), (4, 1, 64), (4, 2, 32), (4, 2, 64), (4, 3, 64), (4, 4, 64), (8, 1, 64), (8, 2, 64), (8, 3, 64),
              (8, 4, 64), (8, 3, 32)]                                     # MISO_FUSED_SHAPES (sdf_fused.hip)
    from miso_amd import _lib
    lib = _lib.load()
    for C, L, H in shapes:
        assert ops.sdf_train_lds_bytes(C, L, H, scat=True) <= ops.LDS_PER_WORKGROUP, (C, L, H)
        #  ... and the formula is what the library computes from the kernel's own PackLayout
        # (miso_sdf_train_lds_bytes: what sdf_train_supported asks when it has the decoder) -- it cannot drift unnoticed
        g = _lib.Grid()
        g.n_levels = L
        for a in range(3):
            g.bound_min[a], g.bound_max[a] = -1.0, 1.0
        for l in range(L):
            lv = g.level[l]
            lv.C, lv.X, lv.Y, lv.Z = C, 8, 8, 8
            lv.sC, lv.sX, lv.sY, lv.sZ = 1, C, C * 8, C * 64
            lv.data, lv.grad = None, 0x100000 * (l + 1)
        m = _lib.Mlp()
        m.n_linear, m.in_dim, m.hidden_dim, m.out_dim = 3, C * L, H, 1
        for scat in (True, False):
            want = ops.sdf_train_lds_bytes(C, L, H, scat=scat)
            if not scat:      # nothing scattered: eight wavefronts' tiles beside the pack (the four-wavefront form is smaller)
                want = max(want, want + 4 * 4 * 64 * ((C * L + 3) // 4 * 4 + 4))
            assert lib.miso_sdf_train_lds_bytes(ctypes.byref(g), ctypes.byref(m), int(scat)) == want, (C, L, H, scat)
    m.hidden_dim = 48
    assert lib.miso_sdf_train_lds_bytes(ctypes.byref(g), ctypes.byref(m), 1) == 0          # a shape outside the table
    # (round 6: the bf16x3 pack -- 18 432 dwords of matrix pieces + 196 of biases / output weights at 64 x 64 with F in 17..32)
    assert ops.sdf_train_lds_bytes(8, 3, 64, scat=False) == 4 * (18628 + 4 * 64 * 28)       # four-wavefront form: 101.5 KB
    assert ops.sdf_train_lds_bytes(8, 4, 64, scat=True) == 4 * (18628 + 4 * (64 * 36 + 64 * 4 * 8))
    feats = [torch.zeros(1, 8, 4, 4, 4)] * 3
    meta = ops.GridMeta((-1.0,) * 3, (1.0,) * 3, 0, 0)
    assert ops.sdf_train_supported(feats, meta, [feats[0], None, None]) and not ops.sdf_train_supported(feats, meta, [None] * 3)
