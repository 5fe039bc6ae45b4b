"""Pins the CPU restatement (oracle/pbrt_oracle.c) and the host front end against output of the
UNMODIFIED reference binary: tests/golden/*.pfm + *.json were rendered by oracle/_ref/pbrt_oracle
(oracle/make_golden.py).  Bit-exact images and identical ray statistics are required."""
import json
import os

import numpy as np
import pytest

from conftest import GOLD, golden_names


@pytest.mark.parametrize("name", golden_names())
def test_oracle_matches_reference_image_and_stats(pkg, oracle, name):
    scene = pkg.HostScene(os.path.join(GOLD, name + ".pbrt"))
    img, cn = oracle.render_image(scene)
    ref = pkg.read_pfm(os.path.join(GOLD, name + ".pfm"))
    assert img.shape == ref.shape
    assert np.array_equal(img, ref), f"max |diff| {np.abs(img - ref).max()}"
    stats = json.load(open(os.path.join(GOLD, name + ".json")))
    for k, v in stats.items():  # the reference's own STAT_COUNTERs (scene.cpp:40-42, integrator.cpp:48, triangle.cpp:45)
        assert cn[k] == v, k


def test_reference_binary_live_when_present(pkg, oracle, tmp_path):
    """Where oracle/_ref exists (build container and GPU box), re-render one scene with it now."""
    if not os.path.exists(oracle.REF_BINARY):
        pytest.skip("oracle/_ref/pbrt_oracle not built here")
    name = "cornell_40x24"
    out = str(tmp_path / "live.pfm")
    oracle.run_reference(os.path.join(GOLD, name + ".pbrt"), out, nthreads=2)
    assert np.array_equal(pkg.read_pfm(out), pkg.read_pfm(os.path.join(GOLD, name + ".pfm")))
