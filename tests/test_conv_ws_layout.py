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


def test_combine_area_read_is_conflict_free():
    """Epilogue: block (wave, tile, register group q) = 64 x 16 bytes at a 1152-byte pitch, slot 2 * pixel + half; thread t reads
    (pixel t / 8, channel quad t % 8 = 2 q + half).  A 1024-byte pitch would be 2-way conflicted."""
    def worst(pitch):
        w = 1
        for g in GROUPS:
            slots = {}
            for l in g:
                px, c4 = l >> 3, l & 7
                addr = (c4 >> 1) * pitch + (px * 2 + (c4 & 1)) * 16
                slots.setdefault((addr // 16) % 16, set()).add(addr)
            w = max(w, max(len(v) for v in slots.values()))
        return w
    assert worst(1152) == 1
    assert worst(1024) == 2
