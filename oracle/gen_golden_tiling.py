"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Tile-geometry fixture from the REAL reference (/root/reference): what
`VAEHook.split_tiles` (SUPIR/utils/tilevae.py:717-774, with `get_best_tile_size` :702-715) and `_sliding_windows`
(sgm/modules/diffusionmodules/sampling.py:753-766) return over a sweep of ragged sizes -- maps that are not multiples of the tile or the
stride, long thin images, sizes just above the "unnecessary to tile" threshold (:693), encoder (pad 32, output // 8) and decoder (pad 11,
output x 8) hooks.  The reference has no tests of its own for these; the sweep stores one SHA-1 per case (of the repr of the returned
lists) plus a handful of cases in full, so the fixture stays a few kilobytes.

    python -m oracle.gen_golden_tiling      # seconds; writes tests/golden/golden_tiling.json
"""
import contextlib
import hashlib
import io
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_import as R  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "golden_tiling.json")


def vae_cases():
    """(h, w, tile_size, is_decoder): the encoder sees pixels, the decoder latents."""
    cases = []
    for dec, sizes, tiles in ((False, (65, 96, 129, 577, 600, 1000, 1024, 1111, 2048, 4096), (256, 512, 1024)),
                              (True, (23, 24, 64, 75, 86, 100, 128, 139, 256, 512), (32, 64, 128))):
        for ts in tiles:
            for h in sizes:
                for w in sizes:
                    cases.append((h, w, ts, dec))
    return cases


def window_cases():
    return [(h, w, t, s) for t, s in ((128, 64), (16, 8), (128, 96), (64, 64), (96, 40)) for h in (t, t + 1, t + s - 1, t + s, 2 * t + 7, 512)
            for w in (t, t + 3, t + s, 3 * t - 5, 640)]


def digest(obj):
    return hashlib.sha1(repr(obj).encode()).hexdigest()


def main():
    R.load_reference()
    import sgm.modules.diffusionmodules.sampling as S
    from SUPIR.utils.tilevae import VAEHook
    gold = {"vae": {}, "windows": {}, "full": {}}
    for (h, w, ts, dec) in vae_cases():
        hook = VAEHook(None, ts, dec, False, False, False)
        with contextlib.redirect_stdout(io.StringIO()):
            inb, outb = hook.split_tiles(h, w)
        key = f"{h},{w},{ts},{int(dec)}"
        gold["vae"][key] = digest(([list(map(int, b)) for b in inb], [list(map(int, b)) for b in outb]))
        if (h, w, ts, dec) in ((1111, 600, 512, False), (139, 75, 64, True)):
            gold["full"][key] = [[list(map(int, b)) for b in inb], [list(map(int, b)) for b in outb]]
    for (h, w, t, s) in window_cases():
        gold["windows"][f"{h},{w},{t},{s}"] = digest([tuple(map(int, c)) for c in S._sliding_windows(h, w, t, s)])
    json.dump(gold, open(OUT, "w"), indent=0, sort_keys=True)
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", len(gold["vae"]), "vae cases,", len(gold["windows"]), "window cases")


if __name__ == "__main__":
    main()
