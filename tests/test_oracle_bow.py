"""Oracle checks for DBoW3::Vocabulary::transform (reference thirdParty/DBow3/src/Vocabulary.cpp:752-874, BowVector.cpp:31-88,
FeatureVector.cpp:31-45).  Parity unpinned: the reference's vocabulary file is a missing blob and DBoW3 needs OpenCV, so the oracle
(oracle/bow_oracle.cpp) is checked against a dict-based Python statement of the same text on synthetic trees."""
import math

import numpy as np
import pytest

from ydorbslam_amd.synth import synth_vocabulary


def _descriptors(tree, n, seed):
    """Leaf descriptors with noise: every descent is decided by real distances, several features share a word."""
    rng = np.random.default_rng(seed)
    leaves = np.flatnonzero(np.diff(tree["child_begin"]) == 0)
    pick = rng.choice(leaves, n)
    d = tree["node_desc"][pick].copy()
    flips = rng.random((n, 256)) < 0.08
    return np.packbits(np.unpackbits(d, axis=1) ^ flips.astype(np.uint8), axis=1)


def _python_transform(tree, desc, levelsup, weighting, norm):
    cb, ci, nd, nw, wd, L = (tree[k] for k in ("child_begin", "child_ids", "node_desc", "node_weight", "node_word", "levels"))
    v, fv, status = {}, {}, 0
    for f, feat in enumerate(desc):
        node, level, nid = 0, 0, (0 if L - levelsup <= 0 else None)
        while True:
            level += 1
            best = None
            for c in ci[cb[node]:cb[node + 1]] if level > 0 else []:
                dist = int(np.unpackbits(feat ^ nd[c]).sum())
                if best is None or dist < best[0]:
                    best = (dist, int(c))
            node = best[1]
            if level == L - levelsup:
                nid = node
            if cb[node + 1] == cb[node]:
                break
        if nid is None:
            nid, status = node, status | 1
        w, wid = float(nw[node]), int(wd[node])
        if w > 0:
            if weighting in (0, 1):
                v[wid] = v[wid] + w if wid in v else w
            elif wid not in v:
                v[wid] = w
            fv.setdefault(nid, []).append(f)
    words = sorted(v)
    vals = [v[w] for w in words]
    if weighting in (0, 1) and vals and norm == 0:
        vals = [x / float(len(vals)) for x in vals]
    if norm:
        s = 0.0
        for x in vals:
            s += abs(x) if norm == 1 else x * x
        if norm == 2:
            s = math.sqrt(s)
        if s > 0:
            vals = [x / s for x in vals]
    nodes = sorted(fv)
    return words, vals, nodes, [fv[k] for k in nodes], status


@pytest.mark.parametrize("weighting,norm", [(0, 1), (1, 2), (2, 1), (3, 0), (0, 0)])
def test_oracle_matches_python_statement(oracle_lib, weighting, norm):
    tree = synth_vocabulary(5, 4, seed=weighting, early_leaf_frac=0.15 if weighting == 0 else 0.0)
    desc = _descriptors(tree, 300, 7 + weighting)
    for levelsup in (2, 4):
        bw, bv, fn, fs, ff, st = oracle_lib.bow_transform(tree, desc, levelsup, weighting, norm)
        words, vals, nodes, lists, pst = _python_transform(tree, desc, levelsup, weighting, norm)
        assert list(bw) == words and st == pst
        assert np.array_equal(bv.view(np.uint64), np.array(vals, np.float64).view(np.uint64))
        assert list(fn) == nodes
        assert [list(ff[fs[i]:fs[i + 1]]) for i in range(len(nodes))] == lists
    assert len(words) < 300 and any(len(l) > 1 for l in lists)


def test_l1_normalised_vector_sums_to_one_and_stopped_words_are_dropped(oracle_lib):
    tree = synth_vocabulary(6, 3, seed=9, stopped_frac=0.3)
    desc = _descriptors(tree, 400, 1)
    bw, bv, fn, fs, ff, st = oracle_lib.bow_transform(tree, desc, 2, 0, 1)
    assert abs(bv.sum() - 1.0) < 1e-12 and st == 0
    stopped = set(tree["node_word"][(tree["node_weight"] == 0) & (np.diff(tree["child_begin"]) == 0)])
    assert not stopped & set(bw) and len(ff) < 400 and len(set(ff)) == len(ff)
