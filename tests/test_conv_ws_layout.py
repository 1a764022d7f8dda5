"""CPU check of the LDS patch layout of convws_kernel (diffusiontexturepainting_amd/csrc/conv_ws.hip): with an ODD patch pitch
(TW + 3 pixel slots) and the 16-byte chunk c of pixel slot (hy, hx) stored at chunk c ^ (hx & 7), every ds_read_b128 lane group of every
B-fragment read (any wave quarter, pixel tile, tap) touches 16 distinct 16-byte slots of the 256-byte bank row -- no bank conflicts.
Lane groups of ds_read_b128 on gfx950: /opt/skills/guides/MI355X_MICROARCH.md, LDS table."""
import pytest

GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
GROUPS += [[l + 32 for l in g] for g in GROUPS]


def worst_conflict(th, tw, ni, pw):
    hp1 = (th + 2) * pw
    rpt = 32 // tw
    tpi = th // rpt
    tm = ni * th * tw // 32
    worst = 1
    for wave in range(4):
        for j in range(tm):
            for ky in range(3):
                for kx in range(3):
                    for g in GROUPS:
                        slots = {}
                        for l in g:
                            half, fr = l >> 5, l & 31
                            px, pyl = fr % tw, fr // tw
                            hy, hx = rpt * (j % tpi) + pyl + ky, px + kx
                            slot = (j // tpi) * hp1 + hy * pw + hx
                            addr = slot * 128 + (((2 * wave + half) ^ (hx & 7)) << 4)
                            slots.setdefault((addr // 16) % 16, set()).add(addr)
                        worst = max(worst, max(len(v) for v in slots.values()))
    return worst


@pytest.mark.parametrize("th,tw,ni", [(8, 8, 3), (16, 16, 1)])
def test_odd_pitch_patch_is_conflict_free(th, tw, ni):
    assert worst_conflict(th, tw, ni, tw + 3) == 1
    assert worst_conflict(th, tw, ni, tw + 2) == 2  # the natural (even) halo pitch is 2-way conflicted: why the pitch is padded


WRITE_GROUPS = [list(range(8 * i, 8 * i + 8)) for i in range(8)]  # ds_write_b128: eight groups of eight consecutive lanes, bank = (a / 4) mod 32


def _worst(groups, addr_of, bank_row_bytes):
    w = 1
    for g in groups:
        banks = {}
        for l in g:
            a = addr_of(l)
            for b in range(a, a + 16, 4):
                banks.setdefault((b // 4) % (bank_row_bytes // 4), set()).add(a)
        w = max(w, max(len(v) for v in banks.values()))
    return w


def test_combine_area_is_conflict_free_on_both_sides():
    """Epilogue: block (wave, tile, register group q) = 64 x 16 bytes at a 1152-byte pitch.  Every wave WRITES its accumulators with
    ds_write_b128 (lane = (pixel, half): groups of 8 consecutive lanes, 32 banks) and thread t of the workgroup READS (pixel t / 8, channel quad
    t % 8 = 2 q + half) with ds_read_b128 (16-lane groups, 64 banks).  Round 6 (the round-5 verdict's bank-conflict item): the slot
    2 pixel + half of rounds 4-5 was conflict-free for the reads only -- the writes were 2-way conflicted, which is every conflict cycle the SQ
    counter showed for convws_kernel; slot (2 pixel + half) ^ ((pixel >> 2) & 1) is conflict-free for both."""
    old = lambda px, h: 2 * px + h
    new = lambda px, h: (2 * px + h) ^ ((px >> 2) & 1)
    assert len({new(px, h) for px in range(32) for h in range(2)}) == 64  # still a permutation of the block's 64 slots

    def write_worst(slot):
        return _worst(WRITE_GROUPS, lambda l: slot(l & 31, l >> 5) * 16, 128)

    def read_worst(slot, pitch):
        return max(_worst(GROUPS, lambda l, w=w: ((l & 7) >> 1) * pitch + slot((w * 64 + l) >> 3, l & 1) * 16, 256) for w in range(4))

    assert write_worst(old) == 2 and read_worst(old, 1152) == 1   # rounds 4-5
    assert write_worst(new) == 1 and read_worst(new, 1152) == 1   # round 6
    assert read_worst(new, 1024) == 2                             # the pitch still matters
