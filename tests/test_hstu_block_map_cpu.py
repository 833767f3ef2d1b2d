"""The block -> (sequence, head, rank) maps of the attention kernels (`seq_head_of_block`, csrc/hstu_attn.hip), restated with
Python integers: whatever map a launch takes -- the plain (H, B, blocks) grid, the rotation of jagged batches or the
column-major map of dense long batches -- every (sequence, head, rank) triple must be produced by exactly ONE block of the
grid, or a block's worth of rows would silently stay unwritten; and the properties the maps exist for must hold: the
rotation deals a column's blocks over all XCDs, the column-major map keeps a column on one.  (The kernels themselves are
checked on the GPU: tests/test_hstu_gpu.py::test_dense_long_batch_block_map_equals_the_jagged_one, the C4 test.)"""
import itertools

import pytest


def block_map(x, y, z, H, B, nz, rot, colmajor, dense):
    """(b, h, rank) of block (x, y, z) of a grid (H, B, nz); mirrors seq_head_of_block statement by statement"""
    b0, h0, z0 = y, x, z
    if rot == 0 and not colmajor:
        return b0, h0, z0
    bh = H * B
    step = -rot if rot < 0 else rot
    lin0 = x + H * y
    slot = (lin0 + step * z) % bh
    b1, h1 = slot // H, slot % H
    lin = lin0 + bh * z
    xcd, k = lin & 7, lin >> 3
    col = xcd + 8 * (k // nz)
    z2 = k % nz
    cm_ok = colmajor and (bh & 7) == 0 and nz >= 16
    if dense:
        return (col // H, col % H, z2) if cm_ok else (b0, h0, z0)
    if rot != 0:
        return b1, h1, z0
    return b0, h0, z0


GRIDS = [(4, 32, 32), (4, 32, 4), (2, 4, 16), (8, 8, 17), (1, 8, 16), (3, 5, 16), (4, 6, 20), (1, 1, 1), (2, 64, 3), (4, 7, 33)]


@pytest.mark.parametrize("H,B,nz", GRIDS)
@pytest.mark.parametrize("dense", [False, True])
@pytest.mark.parametrize("rot_kind", ["default", "off", "always1"])
@pytest.mark.parametrize("colmajor", [0, 1])
def test_every_sequence_head_rank_is_served_by_exactly_one_block(H, B, nz, dense, rot_kind, colmajor):
    rot = {"default": -H, "off": 0, "always1": 1}[rot_kind]
    seen = {}
    for z, y, x in itertools.product(range(nz), range(B), range(H)):
        t = block_map(x, y, z, H, B, nz, rot, colmajor, dense)
        assert 0 <= t[0] < B and 0 <= t[1] < H and 0 <= t[2] < nz
        assert t not in seen, f"{t} from blocks {seen[t]} and {(x, y, z)}"
        seen[t] = (x, y, z)
    assert len(seen) == H * B * nz


def _xcd(x, y, z, H, B):
    return (x + H * (y + B * z)) % 8        # workgroup ids are dealt round-robin over the 8 XCDs, x fastest


def test_rotation_deals_a_column_over_all_xcds_and_the_column_major_map_keeps_it_on_one():
    H, B, nz = 4, 32, 32
    by_col_rot, by_col_cm, by_col_plain = {}, {}, {}
    for z, y, x in itertools.product(range(nz), range(B), range(H)):
        xcd = _xcd(x, y, z, H, B)
        b, h, _ = block_map(x, y, z, H, B, nz, -H, 1, False)
        by_col_rot.setdefault((b, h), set()).add(xcd)
        b, h, _ = block_map(x, y, z, H, B, nz, -H, 1, True)
        by_col_cm.setdefault((b, h), set()).add(xcd)
        by_col_plain.setdefault((y, x), set()).add(xcd)
    assert all(len(s) == 1 for s in by_col_plain.values())      # the plain grid: a column never leaves its XCD ...
    assert all(len(s) == 1 for s in by_col_cm.values())         # ... nor under the column-major map (one column after another)
    assert all(len(s) >= 2 for s in by_col_rot.values())        # the rotation moves it (at H = 4: between the two parities) ...
    # ... so that the blocks of sequences of ONE parity no longer pile up on half of the XCDs: per-XCD block counts of the
    # seven longest sequences of the C4 batch (positions 1, 13, 17, 19, 22, 26, 27) under both maps
    long_seqs = {1, 13, 17, 19, 22, 26, 27}
    load_plain, load_rot = [0] * 8, [0] * 8
    for z, y, x in itertools.product(range(nz), range(B), range(H)):
        xcd = _xcd(x, y, z, H, B)
        if y in long_seqs:
            load_plain[xcd] += nz - z
        if block_map(x, y, z, H, B, nz, -H, 1, False)[0] in long_seqs:
            load_rot[xcd] += nz - z
    assert max(load_plain) > 2.4 * min(load_plain)              # 5 odd against 2 even sequences
    assert max(load_rot) < 1.1 * min(load_rot)
