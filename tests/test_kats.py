"""Known-answer pixels (tests/golden/kats.json, made by tests/golden/make_kats.py from the
reference's formulas in numpy float32 and float64, independently of oracle/ and of the product)
asserted on the CPU oracle and, with `-m gpu`, on the HIP path.  Bar: every stored byte within 1
(north_star's tolerance); the scenes are the reference's eight fixtures (all of its own test
material: each carries a header comment describing the expected picture, which make_kats.py
turns into assertions as well) plus a transmitting pane and a `pp` patch."""
import json
import os

import numpy as np
import pytest

from conftest import FIXTURE_DIR, REFERENCE_FIXTURES, read_nff

with open(os.path.join(FIXTURE_DIR, "kats.json")) as f:
    KATS = json.load(f)
SCENES = sorted({p["scene"] for p in KATS["pixels"]})


def scene_text(name):
    return KATS["scenes"][name].encode() if name in KATS["scenes"] else read_nff(name)


def check(name, u8):
    mine = [p for p in KATS["pixels"] if p["scene"] == name]
    assert len(mine) >= 3
    for p in mine:
        got = u8[p["y"], p["x"]].astype(int)
        assert np.abs(got - np.array(p["rgb"])).max() <= 1, (name, p["x"], p["y"], got.tolist(), p["rgb"], p["what"])


def test_every_reference_fixture_has_known_answers():
    assert set(REFERENCE_FIXTURES) <= set(SCENES)
    assert {"kat-window", "kat-patch"} <= set(SCENES)


@pytest.mark.parametrize("name", SCENES)
def test_oracle_reproduces_known_answers(oracle, name):
    s = oracle.OracleScene(scene_text(name))
    for pow_mode in (oracle.POW_LIBM, oracle.POW_PORTABLE):
        _, u8, _, _ = s.render(max_depth=5, pow_mode=pow_mode, want_f32=False, threads=4)
        check(name, u8)


@pytest.mark.gpu
@pytest.mark.parametrize("name", SCENES)
def test_hip_path_reproduces_known_answers(rtamd, gpu_ctx, name):
    sc = rtamd.Scene.from_nff(gpu_ctx, scene_text(name))
    for fp_mode in (rtamd.FP_EXACT, rtamd.FP_FAST):
        for exhaustive in (False, True):
            _, u8, _ = sc.render(rtamd.RenderOptions(max_depth=5, fp_mode=fp_mode, exhaustive_leaves=exhaustive), want_f32=False)
            check(name, u8)
