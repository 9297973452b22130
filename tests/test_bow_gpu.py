"""GPU parity: ydorb_vocabulary_transform (C ABI) vs oracle/bow_oracle.cpp, the restatement of DBoW3::Vocabulary::transform
(reference thirdParty/DBow3/src/Vocabulary.cpp:752-874).  Bar: identical ids and identical double bit patterns.
Parity unpinned (synthetic trees: the reference's vocabulary file is a missing blob)."""
import numpy as np
import pytest

from test_oracle_bow import _descriptors
from ydorbslam_amd.synth import synth_vocabulary

pytestmark = pytest.mark.gpu

W = {"TF_IDF": 0, "TF": 1, "IDF": 2, "BINARY": 3}
N = {"none": 0, "L1": 1, "L2": 2}


def _check(oracle_lib, tree, voc, descs, levelsup, weighting, norm, expect_status=None):
    got = voc.transform(descs, levelsup)
    for d, (bw, bv, fn, fs, ff, st) in zip(descs, got):
        obw, obv, ofn, ofs, off, ost = oracle_lib.bow_transform(tree, d, levelsup, W[weighting], N[norm])
        assert np.array_equal(bw, obw) and np.array_equal(bv.view(np.uint64), obv.view(np.uint64))
        assert np.array_equal(fn, ofn) and np.array_equal(fs, ofs) and np.array_equal(ff, off)
        assert st == ost
        if expect_status is not None:
            assert st == expect_status


@pytest.mark.parametrize("weighting,norm", [("TF_IDF", "L1"), ("TF", "L2"), ("IDF", "L1"), ("BINARY", "none"), ("TF_IDF", "none")])
def test_transform_matches_the_oracle(oracle_lib, weighting, norm):
    import ydorbslam_amd as y
    tree = synth_vocabulary(10, 4, seed=3)
    voc = y.Vocabulary(tree, weighting, norm)
    descs = [_descriptors(tree, n, 100 + n) for n in (1000, 1, 257, 0, 2048, 777)]
    _check(oracle_lib, tree, voc, descs, 2, weighting, norm, expect_status=0)
    _check(oracle_lib, tree, voc, descs[:2], 4, weighting, norm)   # levelsup >= L: everything under the root


def test_orb_vocabulary_shape_on_extracted_descriptors(oracle_lib):
    """k = 10, L = 6 is the shape of the reference's ORB vocabulary (frame.cpp:269-270); 10^6 words would be 35 MB of node
    descriptors, so the synthetic tree thins out below level 3.  Descriptors come from the extractor."""
    import ydorbslam_amd as y
    from ydorbslam_amd.synth import synth_frame
    tree = synth_vocabulary(10, 6, seed=1, early_leaf_frac=0.55)
    voc = y.Vocabulary(tree)
    ex = y.OrbExtractor(1000, max_batch=3)
    descs = [d for _, d in ex.extract_batch(np.stack([synth_frame(640, 480, 70 + i) for i in range(3)]))]
    _check(oracle_lib, tree, voc, descs, 4, "TF_IDF", "L1")
    got = voc.transform(descs, 4)
    assert all(abs(g[1].sum() - 1.0) < 1e-9 for g in got)


def test_feature_vector_feeds_search_by_bow(oracle_lib):
    """The CSR that comes out is the one ydorb_search_by_bow takes (SURVEY 8f rank 4 -> M5)."""
    import ydorbslam_amd as y
    from helpers import shifted_pair
    tree = synth_vocabulary(8, 4, seed=5)
    voc = y.Vocabulary(tree)
    a, b = shifted_pair(640, 480, 90, 4, -2)
    ex = y.OrbExtractor(800, max_batch=2)
    (ka, da), (kb, db) = ex.extract_batch(np.stack([a, b]))
    ta, tb = voc.transform([da, db], 2)
    fa = y.FeatureVector(ta[2].astype(np.uint32), ta[3], ta[4])
    fb = y.FeatureVector(tb[2].astype(np.uint32), tb[3], tb[4])
    va = np.ones(len(ka), np.uint8)
    n, out = y.OrbMatcher(0.75, True).search_by_bow(3, ka, da, va, fa, kb, db, None, fb)
    on, oout = oracle_lib.search_by_bow(3, ka, da, va, (ta[2].astype(np.uint32), ta[3], ta[4]), kb, db, None, (tb[2].astype(np.uint32), tb[3], tb[4]), 0.75, True)
    assert n == on and np.array_equal(out, oout) and n > 20


def test_argument_checks():
    import ydorbslam_amd as y
    tree = synth_vocabulary(4, 2, seed=0)
    bad = dict(tree)
    bad["child_ids"] = tree["child_ids"].copy()
    bad["child_ids"][1] = bad["child_ids"][0]          # a node with two parents
    with pytest.raises(y.YdorbError):
        y.Vocabulary(bad)
    voc = y.Vocabulary(tree)
    with pytest.raises(y.YdorbError):
        voc.transform([np.zeros((8193, 32), np.uint8)])


def test_one_vocabulary_shared_by_threads(oracle_lib):
    """Tracking, local mapping and loop closing call computeBoW on the same vocabulary (frame.cpp:265-272, keyFrame): concurrent
    transforms on one handle must not disturb each other."""
    import threading
    import ydorbslam_amd as y
    tree = synth_vocabulary(8, 4, seed=11)
    voc = y.Vocabulary(tree)
    sets = [_descriptors(tree, 600 + 50 * i, 200 + i) for i in range(6)]
    want = [oracle_lib.bow_transform(tree, d, 2, 0, 1) for d in sets]
    bad = []

    def work(i):
        for _ in range(10):
            bw, bv, fn, fs, ff, st = voc.transform([sets[i]], 2)[0]
            if not (np.array_equal(bw, want[i][0]) and np.array_equal(bv.view(np.uint64), want[i][1].view(np.uint64)) and np.array_equal(ff, want[i][4])):
                bad.append(i)

    th = [threading.Thread(target=work, args=(i,)) for i in range(6)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not bad
