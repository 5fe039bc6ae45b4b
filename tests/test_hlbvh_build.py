"""HLBVH construction: the device builder pg_hlbvh_build against the host front end's
HLBVHBuild, which the goldens hlbvh_cornell / hlbvh_synthetic pin against the reference (identical images and node-visit
counts).  Bar: bit-identical LinearBVHNode arrays and primitive order."""
import os

import numpy as np
import pytest

from conftest import GOLD


def soup_bounds(n, seed, clustered=False, duplicates=False, flat_axis=None):
    rng = np.random.default_rng(seed)
    c = rng.random((n, 3)).astype(np.float32) * np.float32(100) - np.float32(50)
    if clustered:  # a few dense clusters: long runs of equal high Morton bits, deep treelets
        k = rng.integers(0, 5, n)
        c = (rng.normal(size=(5, 3))[k] * 30 + rng.normal(size=(n, 3)) * 0.01).astype(np.float32)
    if duplicates:  # identical centroids: runs whose 30 Morton bits agree (the bitIndex == -1 leaves)
        c[: n // 2] = c[0]
    if flat_axis is not None:
        c[:, flat_axis] = np.float32(3.25)
    h = (rng.random((n, 3)) * 0.5).astype(np.float32)
    return np.concatenate([c - h, c + h], axis=1).astype(np.float32)


def test_host_builder_invariants(pkg):
    """The host builder on bare bounds: every primitive once, leaves within their nodes' bounds, root = union of all."""
    b = soup_bounds(5000, 1)
    nodes, order = pkg.hlbvh_build(b, 4, device=False)
    assert sorted(order.tolist()) == list(range(len(b)))
    assert np.array_equal(nodes["bmin"][0], b[:, :3].min(axis=0)) and np.array_equal(nodes["bmax"][0], b[:, 3:].max(axis=0))
    leaves = nodes[nodes["nprims"] > 0]
    assert leaves["nprims"].sum() == len(b)
    for lf in leaves[:: max(1, len(leaves) // 200)]:
        pb = b[order[lf["offset"]: lf["offset"] + lf["nprims"]]]
        assert np.array_equal(lf["bmin"], pb[:, :3].min(axis=0)) and np.array_equal(lf["bmax"], pb[:, 3:].max(axis=0))


CASES = [(1, 4, {}), (2, 4, {}), (3, 1, {}), (7, 4, {}), (100, 1, {}), (1000, 4, {}), (1000, 255, {}), (4096, 2, dict(clustered=True)),
         (5000, 4, dict(duplicates=True)), (20000, 4, dict(flat_axis=1)), (50000, 3, dict(clustered=True, duplicates=True)), (300000, 4, {}),
         (1000000, 4, dict(clustered=True))]


@pytest.mark.gpu
@pytest.mark.parametrize("n,max_prims,kw", CASES)
def test_device_build_equals_host_build(gpu, n, max_prims, kw):
    b = soup_bounds(n, 17 + n, **kw)
    hn, ho = gpu.hlbvh_build(b, max_prims, device=False)
    dn, do = gpu.hlbvh_build(b, max_prims, device=True)
    assert np.array_equal(ho, do)
    assert len(hn) == len(dn)
    assert hn.tobytes() == dn.tobytes(), f"first differing node {np.flatnonzero(hn != dn)[:4]}"


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["hlbvh_cornell", "hlbvh_synthetic"])
def test_scene_with_device_built_bvh(gpu, name):
    """The front end with --devicebvh: same flattened scene (nodes, primitive order) and therefore the same image."""
    host = gpu.HostScene(os.path.join(GOLD, name + ".pbrt"))
    gpu.set_device_bvh(True)
    try:
        dev = gpu.HostScene(os.path.join(GOLD, name + ".pbrt"))
    finally:
        gpu.set_device_bvh(False)
    assert host.nodes().tobytes() == dev.nodes().tobytes()
    assert np.array_equal(host.indices(), dev.indices())
    img, _ = gpu.render_scene(dev)
    assert np.array_equal(img, gpu.render_scene(host)[0])
