#!/usr/bin/env python3
"""SHA-256 of the space partitions of the largest benchmark scenes as the CPU ORACLE's own O(n^2)
builder makes them (oracle/ref_scene.cpp, a restatement of the reference's
src/space_partition.rs:40-258), so that the GPU tests can hold the product's partition -- whose big
nodes are costed on the device -- to a tree nobody in the product had a hand in, before the oracle
adopts that partition for its renders of the same scenes.

    python tests/golden/make_tree_hashes.py [threads]      # rewrites tests/golden/tree_hashes.json

Minutes of CPU for mount size 8 (131,076 primitives).  The hash covers the canonical preorder dump
(node, under subtree, over subtree): per node the box {corner, extent} as f32 bits, the axis
(-1 = leaf), the plane's bits, the leaf's size, then every leaf list in that order.
"""
import hashlib
import importlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

# (key, generator, its arguments): the scenes of tests/test_gpu_fullsize.py that are too big for the
# oracle to build inside a test
SCENES = [
    ("mount8_lights8", "mount", dict(size=8, lights=8, width=64, height=64)),
    ("tetra6", "tetra", dict(size=6, width=64, height=64)),
    ("rings6", "rings", dict(size=6, width=64, height=64)),
]


def tree_hash(boxes, axis, plane, counts, refs):
    """The one definition of the hash, for the oracle's dump and the library's alike."""
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(boxes, dtype=np.float32).tobytes())
    h.update(np.ascontiguousarray(axis, dtype=np.int32).tobytes())
    h.update(np.ascontiguousarray(plane, dtype=np.float32).tobytes())
    h.update(np.ascontiguousarray(counts, dtype=np.uint32).tobytes())
    h.update(np.ascontiguousarray(refs, dtype=np.uint32).tobytes())
    return h.hexdigest()


def main():
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else (os.cpu_count() or 1)
    spd = importlib.import_module("rust-raytracer_amd.spd")  # (scene text only: no kernels, no builder)
    from oracle import oracle
    out = {"made_by": "tests/golden/make_tree_hashes.py", "builder": "oracle (CPU restatement of src/space_partition.rs:40-258)", "scenes": {}}
    for key, gen, kw in SCENES:
        text = spd.GENERATORS[gen](**kw)
        t0 = time.time()
        o = oracle.OracleScene(text, build_threads=threads)
        t = o.tree()
        out["scenes"][key] = {
            "generator": gen, "args": {k: v for k, v in kw.items() if k not in ("width", "height")},
            "primitives": int(o.info.n_prims), "nodes": int(o.info.nodes), "leaf_refs": int(o.info.leaf_refs),
            "max_depth": int(o.info.max_depth), "max_leaf": int(o.info.max_leaf),
            "sha256": tree_hash(t["boxes"], t["axis"], t["plane"], t["count"], t["refs"]),
        }
        print(key, out["scenes"][key], "%.1f s" % (time.time() - t0), flush=True)
    with open(os.path.join(HERE, "tree_hashes.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
