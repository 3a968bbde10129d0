#!/usr/bin/env python3
"""Exhaustive check of a conflict-free LDS layout for conv_tile16 (measured, NOT adopted: it removes the bank conflicts -- PMC
SQ_LDS_BANK_CONFLICT 294912 -> 0 at (2,16,32^3) -- but its per-read address arithmetic replaces immediate offsets and the kernel got
slower, 13.4 -> 14.4 us; the LDS is 7 % busy in this kernel, so the 2-way conflicts of the natural layout cost less).  The layout: voxel records of four 16-byte channel quads, quad q of voxel v stored at slot
q ^ 2*bit2(v); lane (m = lane & 15, kk = lane >> 4) of a ds_read_b128 reads quad kk of voxel v0 + m.  For every alignment v0 and
each of the four 16-lane groups the instruction is served in (MI355X_MICROARCH.md, LDS table) the 16 lanes must cover all sixteen
16-byte slots of the 256-byte bank row."""
GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
          list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)), list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]
def slots(v0, lanes, swizzle):
    out = set()
    for l in lanes:
        v, kk = v0 + (l & 15), l >> 4
        p = kk ^ ((v >> 1) & 2) if swizzle else kk
        out.add((4 * v + p) % 16)
    return out
for sw in (False, True):
    worst = min(len(slots(v0, g, sw)) for v0 in range(64) for g in GROUPS)
    print("swizzle" if sw else "natural", "-> distinct slots per group (16 = conflict-free):", worst)
assert min(len(slots(v0, g, True)) for v0 in range(64) for g in GROUPS) == 16
