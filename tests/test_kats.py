"""Known-answer pixels and known-answer space partitions (tests/golden/kats.json, made by
tests/golden/make_kats.py from the reference's formulas in numpy float32 and float64, independently
of oracle/ and of the product) asserted on the CPU oracle and, with `-m gpu`, on the HIP path.
Bar: every stored byte within 1 (north_star's tolerance); partitions bit for bit.  The scenes are
the reference's eight fixtures (all of its own test material: each carries a header comment
describing the expected picture, which make_kats.py turns into assertions as well), the added
fixtures whose partitions have interior nodes (one with polygons lying on split planes), scenes
under Blinn-Phong shading, a transmitting pane, a `pp` patch, a total internal reflection, a
one-line cylinder under a coloured light, and a strip of fifteen primitives of every kind."""
import json
import os

import numpy as np
import pytest

from conftest import FIXTURE_DIR, REFERENCE_FIXTURES, read_nff
from test_host import canonical

with open(os.path.join(FIXTURE_DIR, "kats.json")) as f:
    KATS = json.load(f)
# (scene, rendered with --blinn-phong)
CASES = sorted({(p["scene"], bool(p["blinn"])) for p in KATS["pixels"]})
CASE_IDS = [name + ("+blinn" if blinn else "") for name, blinn in CASES]


def scene_text(name):
    return KATS["scenes"][name].encode() if name in KATS["scenes"] else read_nff(name)


def check(case, u8):
    mine = [p for p in KATS["pixels"] if (p["scene"], bool(p["blinn"])) == case]
    assert len(mine) >= 3
    for p in mine:
        got = u8[p["y"], p["x"]].astype(int)
        assert np.abs(got - np.array(p["rgb"])).max() <= 1, (case, p["x"], p["y"], got.tolist(), p["rgb"], p["what"])


def stored_tree(name):
    t = KATS["trees"][name]
    as_f32 = lambda bits: np.array(bits, dtype=np.uint32).view(np.float32)
    return (as_f32(t["boxes"]).reshape(-1, 6), np.array(t["axis"], np.int32), as_f32(t["plane"]),
            np.array(t["count"], np.uint32), np.array(t["refs"], np.uint32))


def same_tree(got, want, what):
    boxes, axis, plane, counts, refs = got
    w_boxes, w_axis, w_plane, w_counts, w_refs = want
    assert np.array_equal(axis, w_axis), what
    assert np.array_equal(plane.view(np.uint32), w_plane.view(np.uint32)), what
    assert np.array_equal(np.ascontiguousarray(boxes, np.float32).view(np.uint32), w_boxes.view(np.uint32)), what
    assert np.array_equal(counts, w_counts), what
    assert np.array_equal(refs, w_refs), what


def test_what_the_known_answers_cover():
    names = {name for name, _ in CASES}
    assert set(REFERENCE_FIXTURES) <= names
    assert {"kat-window", "kat-patch", "kat-tir", "kat-cylinder", "kat-mixed", "extra-onplane", "extra-glass", "extra-cones"} <= names
    assert len(KATS["pixels"]) >= 100 and len(CASES) >= 14
    assert sum(1 for _, blinn in CASES if blinn) >= 2
    multi_node = [n for n in KATS["trees"] if any(a >= 0 for a in KATS["trees"][n]["axis"])]
    assert len(multi_node) >= 2 and "extra-onplane" in multi_node and "kat-mixed" in multi_node
    # the fixture with polygons on split planes: a primitive that the appraisal counted on one side only
    # (src/space_partition.rs:111, :120) sits in both children's lists (:163, :166)
    t = KATS["trees"]["extra-onplane"]
    assert sum(t["count"]) > 13


@pytest.mark.parametrize("name", sorted(KATS["trees"]))
def test_builders_reproduce_known_partitions(rtamd, oracle, name):
    """The product's host builder and the oracle's against the partition make_kats.py derived from
    src/space_partition.rs:40-258 on its own: boxes, axes, planes and leaf lists, bit for bit."""
    want = stored_tree(name)
    text = scene_text(name)
    _, nodes, refs = rtamd.ParsedScene(text).build_tree(threads=2)
    same_tree(canonical(nodes, refs), want, name + ": product")
    t = oracle.OracleScene(text).tree()
    same_tree((t["boxes"], t["axis"], t["plane"], t["count"], t["refs"]), want, name + ": oracle")


@pytest.mark.parametrize("case", CASES, ids=CASE_IDS)
def test_oracle_reproduces_known_answers(oracle, case):
    s = oracle.OracleScene(scene_text(case[0]), use_phong=not case[1])
    for pow_mode in (oracle.POW_LIBM, oracle.POW_PORTABLE):
        _, u8, _, _ = s.render(max_depth=5, pow_mode=pow_mode, want_f32=False, threads=4)
        check(case, u8)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(KATS["trees"]))
def test_device_scene_keeps_the_known_partition(rtamd, gpu_ctx, name):
    sc = rtamd.Scene.from_nff(gpu_ctx, scene_text(name))
    nodes, refs = sc.tree()
    same_tree(canonical(nodes, refs), stored_tree(name), name)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=CASE_IDS)
def test_hip_path_reproduces_known_answers(rtamd, gpu_ctx, case):
    sc = rtamd.Scene.from_nff(gpu_ctx, scene_text(case[0]), use_phong=not case[1])
    for fp_mode in (rtamd.FP_EXACT, rtamd.FP_FAST):
        for exhaustive in (False, True):
            _, u8, _ = sc.render(rtamd.RenderOptions(max_depth=5, fp_mode=fp_mode, exhaustive_leaves=exhaustive), want_f32=False)
            check(case, u8)
