"""Brute-force bank-conflict check of the LDS images the conv kernels read with ds_read_b128.

Model (MI355X_MICROARCH.md, LDS): a ds_read_b128 wave instruction is serviced in four 16-lane groups
{0-3,12-15,20-27}, {4-11,16-19,28-31}, {32-35,44-47,52-59}, {36-43,48-51,60-63}; the bank of byte address a is
(a / 4) % 64, i.e. a 16-byte access occupies one of the 16 slots (a / 16) % 16 of the 256-byte bank row.  A group is
conflict free when its 16 lanes hit 16 different slots (identical addresses would broadcast; none occur here).
"""
GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
          list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
          list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
          list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]


def worst_way(addr_of_lane):
    """max over lane groups of the number of lanes sharing one 16-byte slot."""
    worst = 0
    for grp in GROUPS:
        slots = {}
        for lane in grp:
            slots.setdefault((addr_of_lane(lane) // 16) % 16, set()).add(addr_of_lane(lane))
        worst = max(worst, max(len(v) for v in slots.values()))
    return worst


def n16_fragment_addr(lane, row_base, kk):
    """conv_n16.hip: 128-byte rows (64 x 16-bit), row = row_base + (lane & 15), k-chunk (lane >> 4) + 4 * kk,
    slot = chunk ^ ((row >> 1) & 7)."""
    row = row_base + (lane & 15)
    chunk = (lane >> 4) + 4 * kk
    return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4)


def b3_fragment_addr(lane, row_base):
    """conv_b3.hip (dma16): 64-byte rows (32 bf16), slot = chunk ^ F[(row >> 2) & 3], F = {0, 2, 3, 1}."""
    row = row_base + (lane & 15)
    f = (0x78 >> (2 * ((row >> 2) & 3))) & 3
    return row * 64 + (((lane >> 4) ^ f) << 4)


def n16_dma_image_is_a_permutation(rows=256):
    """The DMA writes LDS lane-linearly (piece = 8 rows, lane -> row l / 8, slot l % 8) and fetches source chunk
    slot ^ ((row >> 1) & 7): every (row, chunk) must land exactly once, at the address the fragment read expects."""
    seen = {}
    for piece in range(rows // 8):
        for lane in range(64):
            row = piece * 8 + (lane >> 3)
            slot = lane & 7
            chunk = slot ^ ((row >> 1) & 7)
            seen[(row, chunk)] = piece * 1024 + lane * 16
    if len(seen) != rows * 8:
        return False
    return all(addr == row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4) for (row, chunk), addr in seen.items())


# conv_n16_patch_kernel: the input window of a 16 x 16 output patch is 18 x 18 pixels of 128 bytes (row = wy * 18 + wx); the
# fragment of output row r under tap (kh, kw) is the 16 consecutive window pixels (r + kh) * 18 + kw + (lane & 15).  18 is
# even, so the bank-row half of a pixel is the parity of its window COLUMN wx, and the swizzle is a function of wx alone:
# slot = chunk ^ PATCH_F[wx], found by exhaustive search so that the reads are conflict free for kw = 0, 1 and 2.
PATCH_F = [0, 0, 1, 1, 2, 2, 4, 4, 5, 5, 6, 6, 2, 2, 6, 6, 0, 0]


def patch_fragment_addr(lane, out_row, kh, kw, kk):
    wx = kw + (lane & 15)
    row = (out_row + kh) * 18 + wx
    chunk = (lane >> 4) + 4 * kk
    return row * 128 + ((chunk ^ PATCH_F[wx]) << 4)


def patch_table_constant():
    v = 0
    for i, f in enumerate(PATCH_F):
        v |= f << (3 * i)
    return v


# conv_b3_patch_kernel: the same window with 64-byte rows (32 bf16 per plane): four rows per bank row, slot = chunk ^
# B3_PATCH_F[wx] (exhaustive search; the bank-row quarter of a pixel is (2 * wy + wx) & 3).
B3_PATCH_F = [0, 0, 0, 0, 2, 2, 0, 0, 0, 0, 2, 2, 2, 2, 2, 2, 0, 0]


def b3_patch_fragment_addr(lane, out_row, kh, kw):
    wx = kw + (lane & 15)
    row = (out_row + kh) * 18 + wx
    return row * 64 + (((lane >> 4) ^ B3_PATCH_F[wx]) << 4)


# conv_b3_win_kernel / conv_n16_win_kernel: a contiguous (1-D) window of flattened pixels; a fragment is 16 consecutive window
# rows starting at ANY row (tap shift (kh-1) * W + (kw-1) for arbitrary W).  Exact search (pairwise-constraint DFS over
# period-16 tables) shows that one table serves every alignment: 64-byte rows slot = chunk ^ ((row & 4) >> 1), 128-byte rows
# slot = chunk ^ (row & 6).
def b3_win_fragment_addr(lane, first_row):
    row = first_row + (lane & 15)
    return row * 64 + (((lane >> 4) ^ ((row & 4) >> 1)) << 4)


def n16_win_fragment_addr(lane, first_row, kk):
    row = first_row + (lane & 15)
    return row * 128 + ((((lane >> 4) + 4 * kk) ^ (row & 6)) << 4)


def main():
    ok = True
    for base in range(0, 256, 16):          # fragment tiles start at multiples of 16 rows
        for kk in (0, 1):
            w = worst_way(lambda l: n16_fragment_addr(l, base, kk))
            ok &= w == 1
    print("conv_n16 fragment reads conflict free:", ok)
    okb = all(worst_way(lambda l: b3_fragment_addr(l, base)) == 1 for base in range(0, 256, 16))
    print("conv_b3 fragment reads conflict free:", okb)
    print("conv_n16 DMA image is the permutation the reads expect:", n16_dma_image_is_a_permutation())
    okp = all(worst_way(lambda l: patch_fragment_addr(l, r, kh, kw, kk)) == 1
              for r in range(16) for kh in range(3) for kw in range(3) for kk in (0, 1))
    print("conv_n16 patch-window fragment reads conflict free (all rows, taps):", okp, hex(patch_table_constant()))
    okw = all(worst_way(lambda l: b3_win_fragment_addr(l, r)) == 1 for r in range(64)) and \
        all(worst_way(lambda l: n16_win_fragment_addr(l, r, kk)) == 1 for r in range(64) for kk in (0, 1))
    print("1-D window fragment reads conflict free at every alignment (b3 and n16):", okw)
    okq = all(worst_way(lambda l: b3_patch_fragment_addr(l, r, kh, kw)) == 1 for r in range(16) for kh in range(3) for kw in range(3))
    print("conv_b3 patch-window fragment reads conflict free (all rows, taps):", okq,
          hex(sum(f << (2 * i) for i, f in enumerate(B3_PATCH_F))))


if __name__ == "__main__":
    main()
