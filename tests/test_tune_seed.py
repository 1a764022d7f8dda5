"""The shipped tune table (diffusiontexturepainting_amd/tune_seed.txt) must belong to THIS build: same key prefix as the engine
writes (a table with an older prefix is silently ignored and every process re-tunes ~500 shapes: round 2 shipped one), tile ids the
launcher knows, and the shapes of the benchmark's headline configurations.  CPU only."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "diffusiontexturepainting_amd")


def _engine_constants():
    eng = open(os.path.join(PKG, "csrc", "engine.hip")).read()
    prefix = re.search(r'snprintf\(key, sizeof\(key\), "(k\d+\|)%d', eng).group(1)
    common = open(os.path.join(PKG, "csrc", "common.h")).read()
    ntiles = int(re.search(r"constexpr int DTP_TILE_IDS = (\d+);", common).group(1))
    lnlin = int(re.search(r"constexpr int DTP_TILE_LNLIN = (\d+);", common).group(1))
    return prefix, ntiles, lnlin


def _rows():
    rows = [ln.split() for ln in open(os.path.join(PKG, "tune_seed.txt")) if ln.strip()]
    assert all(len(r) == 3 for r in rows)
    return [(r[0], int(r[1]), int(r[2])) for r in rows]


def test_seed_matches_the_engine_key_prefix_and_tile_range():
    prefix, ntiles, _ = _engine_constants()
    rows = _rows()
    assert len(rows) >= 500
    assert all(k.startswith(prefix) for k, _, _ in rows), "tune_seed.txt was written by a build with another key prefix: regenerate it (tools/r03_seed.sh)"
    assert all(0 <= t < ntiles and 1 <= s <= 64 for _, t, s in rows)
    assert len({k for k, _, _ in rows}) == len(rows)  # one entry per shape


def test_seed_covers_the_headline_shapes_and_uses_the_round3_kernels():
    prefix, _, lnlin = _engine_constants()
    rows = _rows()
    keys = {k: (t, s) for k, t, s in rows}

    def has(m, n, k):
        return any(key.startswith(f"{prefix}{m},{n},{k},") for key in keys)

    # batch-1 512^2 stamp (3 guidance branches): level-0 conv, FF1 (GEGLU) of levels 0 / 1, a level-3 conv; batch 8: level-0 conv
    for m, n, k in [(12288, 320, 2880), (12288, 2560, 320), (3072, 5120, 640), (192, 1280, 11520), (98304, 320, 2880)]:
        assert has(m, n, k), (m, n, k)
    tiles = [t for _, t, _ in rows]
    assert tiles.count(lnlin) >= 10       # the activation-stationary Linear is the tuner's choice for the K = 320 / 640 folds
    assert any(t in (48, 49) for t in tiles)  # three-images-per-workgroup halo tiles
    assert any(12 <= t < 16 for t in tiles)
    # halo entries split K in whole channel blocks only: a split factor the launcher cannot realise would be re-tuned on every start
    for key, t, s in rows:
        if (12 <= t < 16 or t in (48, 49)) and s > 1:
            f = key[len(prefix):].split(",")
            nkb = -(-int(f[10]) // 64)  # ldw / 64
            units = -(-nkb // 9)
            ups = -(-units // s)
            assert -(-nkb // (ups * 9)) == s, key
