"""The LDS tile-image layout of csrc/attention6h.hip - [plane][64 rows][64 d] bf16, 128-byte rows, the 16-byte unit c of row r stored at unit
c ^ (2 ((r >> 1) & 3)) - is conflict-free for both of the kernel's access patterns under the banking rules of
/opt/skills/guides/MI355X_MICROARCH.md (section LDS): 64 banks of 4 bytes; `ds_read_b128` is served in four groups of 16 lanes
{0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, {32-35, 44-47, 52-59}, {36-43, 48-51, 60-63}; `ds_read_b64_tr_b16` in two groups of 32 lanes; lanes of
one group conflict when they touch the same bank at different addresses.  The lane -> address maps below are the ones the kernel forms
(offR / offT in attn6h_fwd1_kernel); the un-swizzled layout is checked to FAIL, so the test would notice a vacuous rule."""

B128_GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
               list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)), list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]
TR_GROUPS = [list(range(0, 32)), list(range(32, 64))]


def swz(row, on=True):
    return 2 * ((row >> 1) & 3) if on else 0


def row_fragment_addr(lane, ks, blk16, plane, on=True):
    """byte address of the ds_read_b128 of a row fragment: lane (l15, g) reads unit 4 ks + g of row 16 blk16 + l15"""
    l15, g = lane & 15, lane >> 4
    row = 16 * blk16 + l15
    return plane * 8192 + row * 128 + (((4 * ks + g) ^ swz(row, on)) * 16)


def transposed_addr(lane, db, half, b32, plane, on=True):
    """byte address of the ds_read_b64_tr_b16: lane (l15, g) supplies row 32 b32 + 16 half + 4 g + (l15 >> 2), d = 16 db + 4 (l15 & 3) .. + 3"""
    l15, g = lane & 15, lane >> 4
    row = 32 * b32 + 16 * half + 4 * g + (l15 >> 2)
    unit = (2 * db + ((l15 & 3) >> 1)) ^ swz(row, on)
    return plane * 8192 + row * 128 + unit * 16 + 8 * (l15 & 1)


def conflicts(addrs_by_lane, groups, width):
    """extra LDS cycles: per group, for every bank the number of distinct `width`-byte accesses that touch it, minus one"""
    extra = 0
    for grp in groups:
        banks = {}
        for lane in grp:
            a = addrs_by_lane[lane]
            for b in range(a // 4, (a + width) // 4):
                banks.setdefault(b % 64, set()).add(a)
        extra += max(len(v) for v in banks.values()) - 1
    return extra


def test_row_fragments_are_conflict_free():
    for plane in range(3):
        for blk16 in range(4):
            for ks in range(2):
                addrs = [row_fragment_addr(l, ks, blk16, plane) for l in range(64)]
                assert all(a % 16 == 0 for a in addrs)
                assert conflicts(addrs, B128_GROUPS, 16) == 0, (plane, blk16, ks)
    plain = [row_fragment_addr(l, 0, 0, 0, on=False) for l in range(64)]
    assert conflicts(plain, B128_GROUPS, 16) > 0              # 128-byte rows without the swizzle: 8 lanes per bank quartet


def test_transposed_fragments_are_conflict_free():
    for plane in range(3):
        for b32 in range(2):
            for half in range(2):
                for db in range(4):
                    addrs = [transposed_addr(l, db, half, b32, plane) for l in range(64)]
                    assert conflicts(addrs, TR_GROUPS, 8) == 0, (plane, b32, half, db)
    plain = [transposed_addr(l, 0, 0, 0, 0, on=False) for l in range(64)]
    assert conflicts(plain, TR_GROUPS, 8) > 0


def test_swizzle_is_a_permutation_of_a_row_and_the_fragments_cover_the_tile():
    for row in range(64):
        assert sorted(c ^ swz(row) for c in range(8)) == list(range(8))
    # the 2 k-steps x 4 lane groups of the row fragments read every unit of a row exactly once
    for l15 in range(16):
        seen = sorted(((row_fragment_addr(16 * g + l15, ks, 0, 0) - l15 * 128) // 16) for g in range(4) for ks in range(2))
        assert seen == list(range(8))
    # the transposed reads of a 32-row block (2 halves x 4 d blocks) touch every 8-byte piece of its rows exactly once
    pieces = set()
    for half in range(2):
        for db in range(4):
            for lane in range(64):
                pieces.add(transposed_addr(lane, db, half, 0, 0))
    assert len(pieces) == 2 * 4 * 64 and pieces == set(range(0, 32 * 128, 8))


# ---- the tile image of the one-wavefront head-dim-64 BACKWARD kernels (csrc/attention6.hip: swz1 / img1_off): 32x32x16 MFMA access patterns
def swz1(row):
    return (((row >> 1) & 1) << 2) | ((row >> 3) & 3)


def img1_addr(row, col, plane=0, on=True):
    """byte address of bf16 element `col` of row `row`"""
    if on:
        return plane * 8192 + (row * 64 + ((((col >> 3) ^ swz1(row)) << 3) | (col & 7))) * 2
    return plane * 9216 + (row * 72 + col) * 2          # the padded 144-byte rows the other kernels use


def test_backward_one_wavefront_layout_is_conflict_free_where_the_padded_one_is_not():
    def rx(lane, sub, ks, on=True):                     # row fragment: lane (l31, hh) reads elements 8 hh + 16 ks .. + 7 of row 32 sub + l31
        return img1_addr(32 * sub + (lane & 31), 8 * (lane >> 5) + 16 * ks, on=on)

    def ry(lane, sub, s, dt, half, on=True):            # transposed: row 32 sub + 16 s + 8 half + 4 hh + (l15 >> 2), elements 32 dt + 16 ((lane >> 4) & 1) + 4 (lane & 3) .. + 3
        row = 32 * sub + 16 * s + 8 * half + 4 * (lane >> 5) + ((lane & 15) >> 2)
        return img1_addr(row, 32 * dt + 16 * ((lane >> 4) & 1) + 4 * (lane & 3), on=on)

    def st(tid, i, on=True):                            # staging store: thread tid writes 8 bytes of row idx / 16
        idx = tid + 256 * i
        return img1_addr(idx // 16, 4 * (idx % 16), on=on)

    for sub in range(2):
        for ks in range(4):
            assert conflicts([rx(l, sub, ks) for l in range(64)], B128_GROUPS, 16) == 0, (sub, ks)
        for s in range(2):
            for dt in range(2):
                for half in range(2):
                    assert conflicts([ry(l, sub, s, dt, half) for l in range(64)], TR_GROUPS, 8) == 0, (sub, s, dt, half)
    # ds_write_b64: four groups of 16 contiguous lanes
    w64 = [list(range(16 * g, 16 * g + 16)) for g in range(4)]
    for wave in range(4):
        for i in range(4):
            assert conflicts([st(64 * wave + l, i) for l in range(64)], w64, 8) == 0
    # what it replaces: the transposed reads of the padded layout conflict (the 21-27 % of LDS cycles SQ_LDS_BANK_CONFLICT showed)
    assert conflicts([ry(l, 0, 0, 0, 0, on=False) for l in range(64)], TR_GROUPS, 8) > 0
    # every element of the tile has one place
    seen = {img1_addr(r, c) for r in range(64) for c in range(64)}
    assert len(seen) == 64 * 64 and max(seen) == 8192 - 2
