"""Side-by-side review of two renders: A | B | amplified |A-B| as one PNG, plus RMSE / mismatch statistics.

    python tools/imgdiff.py a.pfm b.pfm --out diff.png [--gain 16]

Inputs: PFM (linear radiance, as vkrt_render writes), PNG/PPM (8-bit display values), or .npy float arrays [H,W,3|4].
Linear inputs are shown through post.frag's gamma 1/2.2; statistics are computed on the stored values.
"""
import argparse
import sys

import numpy as np


def read_image(path):
    """-> (float32 [H,W,3], is_linear)"""
    if path.endswith(".npy"):
        a = np.load(path).astype(np.float32)
        return a[..., :3], True
    if path.endswith(".pfm"):
        with open(path, "rb") as f:
            kind = f.readline().strip()
            w, h = (int(x) for x in f.readline().split())
            scale = float(f.readline())
            ch = 3 if kind == b"PF" else 1
            a = np.frombuffer(f.read(), "<f4" if scale < 0 else ">f4").reshape(h, w, ch)[::-1]  # rows bottom-up
        return np.repeat(a, 3, -1).astype(np.float32) if ch == 1 else a.astype(np.float32), True
    from PIL import Image

    return np.asarray(Image.open(path).convert("RGB"), np.float32) / 255.0, False


def display(a, linear):
    return np.clip(a, 0, None) ** (1 / 2.2) if linear else a


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("a")
    ap.add_argument("b")
    ap.add_argument("--out", default="diff.png")
    ap.add_argument("--gain", type=float, default=16.0, help="amplification of |A-B| in the third panel")
    o = ap.parse_args(argv)
    (a, la), (b, lb) = read_image(o.a), read_image(o.b)
    if a.shape != b.shape:
        sys.exit(f"imgdiff: shapes differ {a.shape} vs {b.shape}")
    d = a - b
    stats = {"rmse": float(np.sqrt(np.mean(d * d))), "max_abs": float(np.abs(d).max()),
             "pixels_differing": float(np.mean(np.any(a != b, axis=-1)))}
    print(" ".join(f"{k}={v:.6g}" for k, v in stats.items()))
    from PIL import Image

    panel = np.concatenate([display(a, la), display(b, lb), np.clip(np.abs(d) * o.gain, 0, 1)], axis=1)
    Image.fromarray((np.clip(panel, 0, 1) * 255 + 0.5).astype(np.uint8), "RGB").save(o.out)
    return stats


if __name__ == "__main__":
    main()
