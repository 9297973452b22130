"""ctypes mirror of the vocabulary part of the C ABI (DBoW3::Vocabulary::transform, Vocabulary.cpp:752-824)."""
import ctypes as C

import numpy as np

from ._lib import YdVocabularyTree, check, lib

WEIGHTING = {"TF_IDF": 0, "TF": 1, "IDF": 2, "BINARY": 3}
NORM = {"none": 0, "L1": 1, "L2": 2}


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Vocabulary:
    """A vocabulary tree resident on the GPU.  tree: dict with child_begin, child_ids, node_desc, node_weight, node_word, levels."""

    def __init__(self, tree, weighting="TF_IDF", norm="L1", device=0):
        self._L = lib()
        self._h = C.c_void_p()
        self.tree = {k: np.ascontiguousarray(v) for k, v in tree.items() if k != "levels"}
        self.levels = int(tree["levels"])
        t = self.tree
        cb = t["child_begin"] = t["child_begin"].astype(np.int32)
        ci = t["child_ids"] = t["child_ids"].astype(np.int32)
        nd = t["node_desc"] = t["node_desc"].astype(np.uint8)
        nw = t["node_weight"] = t["node_weight"].astype(np.float64)
        wd = t["node_word"] = t["node_word"].astype(np.int32)
        self.weighting, self.norm = WEIGHTING[weighting], NORM[norm]
        T = YdVocabularyTree(len(cb) - 1, self.levels, _p(cb), _p(ci), _p(nd), _p(nw), _p(wd), self.weighting, self.norm)
        check(self._L.ydorb_vocabulary_create(C.byref(T), device, C.byref(self._h)))

    def close(self):
        if self._h:
            self._L.ydorb_vocabulary_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def transform(self, descs, levelsup=4):
        """descs: list of [n_i, 32] uint8 arrays (one per frame).  Returns per frame
        (bow_word[int32], bow_value[float64], fv_node[int32], fv_start[int32], fv_feat[int32], status)."""
        descs = [np.ascontiguousarray(d, np.uint8).reshape(-1, 32) for d in descs]
        F = len(descs)
        cap = max(1, max(len(d) for d in descs))
        D = np.zeros((F, cap, 32), np.uint8)
        n = np.zeros(F, np.int32)
        for f, d in enumerate(descs):
            D[f, :len(d)] = d
            n[f] = len(d)
        bw = np.zeros((F, cap), np.int32); bv = np.zeros((F, cap), np.float64); nw = np.zeros(F, np.int32)
        fn = np.zeros((F, cap), np.int32); fs = np.zeros((F, cap + 1), np.int32); ff = np.zeros((F, cap), np.int32); nn = np.zeros(F, np.int32)
        st = np.zeros(F, np.int32)
        check(self._L.ydorb_vocabulary_transform(self._h, _p(D), _p(n), F, cap, int(levelsup), _p(bw), _p(bv), _p(nw), _p(fn), _p(fs), _p(ff),
                                                 _p(nn), _p(st)))
        out = []
        for f in range(F):
            k = nn[f]
            out.append((bw[f, :nw[f]].copy(), bv[f, :nw[f]].copy(), fn[f, :k].copy(), fs[f, :k + 1].copy(), ff[f, :fs[f, k]].copy(), int(st[f])))
        return out
