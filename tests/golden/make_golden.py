"""Regenerates tests/golden/*.ppm.gz from the CPU oracle.

PARITY UNPINNED: the reference ships no golden images and cannot be built here, so
these images are the oracle's own output (libm powf, max depth 5, Phong unless the name
ends in -blinn).  They pin the oracle and the GPU path against regressions; the
independent pins are the analytic known-answer pixels in tests/test_oracle.py.

Usage: python tests/golden/make_golden.py
"""
import gzip
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle  # noqa: E402

CASES = [(n, True) for n in ("test-sphere", "test-cone", "test-square", "test-triangle", "test-side-lit",
                             "test-overlap", "test-shadow-1", "test-shadow-2", "extra-glass", "extra-patch",
                             "extra-cones", "extra-onplane")] + [("extra-glass", False), ("extra-cones", False)]


def main():
    oracle.build()
    for name, phong in CASES:
        text = open(os.path.join(HERE, "nff", name + ".nff"), "rb").read()
        s = oracle.OracleScene(text, use_phong=phong)
        _, u8, _, _ = s.render(max_depth=5, want_f32=False)
        out = os.path.join(HERE, name + ("" if phong else "-blinn") + ".ppm.gz")
        hdr = b"P6\n%d %d\n255\n" % (s.width, s.height)
        with open(out, "wb") as raw, gzip.GzipFile(filename="", mode="wb", fileobj=raw, mtime=0) as f:
            f.write(hdr + u8.tobytes())
        print(out, os.path.getsize(out))


if __name__ == "__main__":
    main()
