"""Scene ingestion: the oracle (oracle/pcg_ref.py) against the UNMODIFIED PCGVoxelGenerator.next_world, and the host-side
pieces of scenedreamer_amd/scene.py (palette, tree ordering) that need no GPU."""
import os
import random

import numpy as np
import pytest
import torch


@pytest.mark.needs_reference
def test_pcg_oracle_equals_reference_next_world(tmp_path):
    """imaginaire.model_utils.pcg_gen.PCGVoxelGenerator.next_world, unchanged, on a world directory written from the
    synthetic maps (cv2.imread is served by Pillow: the image has no OpenCV) == oracle/pcg_ref.next_world."""
    from PIL import Image
    from oracle import pcg_ref
    from oracle import ref_harness as RH
    RH.install("oracle")
    import cv2
    cv2.__dict__["imread"] = lambda path, flag=0: np.asarray(Image.open(path).convert("L"))
    from imaginaire.model_utils.pcg_gen import PCGVoxelGenerator
    S = 160
    height, sem, tree, models = pcg_ref.synthetic_world(S, 5)
    np.save(tmp_path / "heightmap.npy", height)
    Image.fromarray(sem, "L").save(tmp_path / "semanticmap.png")
    Image.fromarray(tree, "L").save(tmp_path / "treemap.png")
    gen = PCGVoxelGenerator(sample_size=S)
    random.seed(11)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        gen.next_world("cpu", str(tmp_path), {"assets": models})
    random.seed(11)
    o = pcg_ref.next_world(height, sem, tree, models)
    assert torch.equal(gen.voxel_t, o["voxel_t"]) and int((gen.voxel_t != 0).sum()) > 0
    assert int(((gen.voxel_t >= 34) & (gen.voxel_t <= 63)).sum()) > 50          # trees were pasted
    assert torch.equal(gen.heightmap, o["heightmap"])
    assert torch.equal(gen.current_height_map, o["current_height_map"])
    assert torch.equal(gen.current_semantic_map, o["current_semantic_map"])
    assert torch.equal(gen.trans_mat, o["trans_mat"])


def test_palette_and_tree_rounds():
    from oracle import pcg_ref
    from scenedreamer_amd import scene
    pal, lut = scene.make_palette([28, 9, 8, 1, 9, 30, 26, 34, 58, 0])
    assert pal[0] == 0 and sorted(pal[1:9].tolist()) == [1, 8, 9, 26, 28, 30, 34, 58] and (pal[9:] == 0).all()
    for i in (1, 8, 9, 26, 28, 30, 34, 58):
        assert pal[lut[i]] == i
    with pytest.raises(RuntimeError):
        scene.make_palette(range(1, 300))
    # tree placement consults the RNG exactly like the reference loop; overlapping trees keep their order across rounds
    height, sem, tree, models = pcg_ref.synthetic_world(200, 3)
    hm = scene.normalise_height_map(height).astype(np.int64) + scene.PAD_NUM
    random.seed(4)
    trees = scene.place_trees(tree, hm, [m.numpy() for m in models], (200, 200))
    assert len(trees) > 20
    dims = [tuple(m.shape) for m in models]
    rounds = scene._rounds(trees, dims)
    assert sorted(k for r in rounds for k in r) == list(range(len(trees)))
    where = {k: i for i, r in enumerate(rounds) for k in r}

    def overlap(a, b):
        (h, x, y, m), (hj, xj, yj, mj) = trees[a], trees[b]
        d, e = dims[m], dims[mj]
        return x < xj + e[1] and xj < x + d[1] and y < yj + e[2] and yj < y + d[2] and h < hj + e[0] and hj < h + d[0]
    n_over = 0
    for a in range(len(trees)):
        for b in range(a):
            if overlap(a, b):
                n_over += 1
                assert where[b] < where[a]
    for r in rounds:                       # trees of one launch are mutually disjoint
        for i, a in enumerate(r):
            assert not any(overlap(a, b) for b in r[:i])
