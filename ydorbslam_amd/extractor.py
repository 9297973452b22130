"""Python mirror of YDORBSLAM::OrbExtractor (reference src/orbExtractor.hpp:31-74) over the C ABI."""
import ctypes as C
import numpy as np

from ._lib import YdExtractorConfig, YdorbError, check, lib

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class OrbExtractor:
    """OrbExtractor(nFeatures, scaleFactor, nLevels, iniThFAST, minThFAST) — orbExtractor.cpp:315."""

    def __init__(self, n_features=1000, scale_factor=1.2, n_levels=8, ini_th=20, min_th=7, device=0, max_batch=1, single_stream=False):
        self._L = lib()
        self._h = C.c_void_p()
        cfg = YdExtractorConfig(n_features, scale_factor, n_levels, ini_th, min_th, device, max_batch, 1 if single_stream else 0)   # YDORB_EXTRACTOR_SINGLE_STREAM
        check(self._L.ydorb_extractor_create(C.byref(cfg), C.byref(self._h)))
        self.n_levels = n_levels
        self.max_keypoints = self._L.ydorb_extractor_max_keypoints(self._h)

    def close(self):
        if getattr(self, "_h", None):
            self._L.ydorb_extractor_destroy(self._h)
            self._h = None

    __del__ = close

    # getters of orbExtractor.hpp:42-49
    def tables(self):
        n = self.n_levels
        sf, isf, sf2, isf2 = (np.zeros(n, np.float32) for _ in range(4))
        per = np.zeros(n, np.int32)
        check(self._L.ydorb_extractor_tables(self._h, _p(sf), _p(isf), _p(sf2), _p(isf2), _p(per)))
        return dict(scale=sf, inv_scale=isf, scale2=sf2, inv_scale2=isf2, per_level=per)

    def extract(self, img):
        """extractAndCompute (orbExtractor.cpp:355): host image in, (keypoints, descriptors[N,32]) out."""
        if img is None or img.size == 0:
            return np.zeros(0, KP_DTYPE), np.zeros((0, 32), np.uint8)
        if img.dtype != np.uint8 or img.ndim != 2:
            raise YdorbError("extract expects a 2-D uint8 image (CV_8UC1, orbExtractor.cpp:361)")
        img = np.ascontiguousarray(img)
        cap = self.max_keypoints
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int32(0)
        check(self._L.ydorb_extract(self._h, _p(img), img.shape[1], img.shape[0], img.strides[0], _p(kps), _p(desc), cap, C.byref(n)))
        return kps[:n.value].copy(), desc[:n.value].copy()

    def extract_batch(self, imgs):
        """imgs: [F,H,W] uint8 host array -> list of (keypoints, descriptors) per frame."""
        imgs = np.ascontiguousarray(imgs, np.uint8)
        F, H, W = imgs.shape
        cap = self.max_keypoints
        kps = np.zeros((F, cap), KP_DTYPE)
        desc = np.zeros((F, cap, 32), np.uint8)
        n = np.zeros(F, np.int32)
        check(self._L.ydorb_extract_batch(self._h, _p(imgs), W, H, imgs.strides[1], imgs.strides[0], F, _p(kps), _p(desc), cap, _p(n)))
        return [(kps[f, :n[f]].copy(), desc[f, :n[f]].copy()) for f in range(F)]

    def extract_batch_device(self, d_img_ptr, w, h, stride, frame_stride, n_frames, d_kps_ptr, d_desc_ptr, cap, d_n_ptr, stream=None):
        """Device-resident asynchronous form: raw HBM addresses (e.g. torch tensor .data_ptr())."""
        check(self._L.ydorb_extract_batch_device(self._h, d_img_ptr, w, h, stride, frame_stride, n_frames, d_kps_ptr, d_desc_ptr,
                                                 cap, d_n_ptr, stream))

    def synchronize(self):
        check(self._L.ydorb_extractor_synchronize(self._h))

    def level_dims(self, level, frame=0):
        w, h, s = C.c_int32(), C.c_int32(), C.c_int32()
        ptr = C.c_void_p()
        check(self._L.ydorb_extractor_pyramid(self._h, frame, level, C.byref(ptr), C.byref(w), C.byref(h), C.byref(s)))
        return w.value, h.value, s.value, ptr.value

    def read_level(self, level, frame=0):
        """m_v_imagePyramid[level] with its 19-px border, as a host array [(h+38),(w+38)]."""
        w, h, _, _ = self.level_dims(level, frame)
        out = np.zeros((h + 38, w + 38), np.uint8)
        check(self._L.ydorb_extractor_read_level(self._h, frame, level, _p(out), out.size))
        return out

    def read_pyramid(self, frame=0):
        """m_v_imagePyramid of one frame, all levels with their 19-px borders, in one device-to-host transfer."""
        outs = []
        for l in range(self.n_levels):
            w, h, _, _ = self.level_dims(l, frame)
            outs.append(np.zeros((h + 38, w + 38), np.uint8))
        ptrs = (C.c_void_p * len(outs))(*[o.ctypes.data for o in outs])
        sizes = (C.c_size_t * len(outs))(*[o.size for o in outs])
        check(self._L.ydorb_extractor_read_pyramid(self._h, frame, ptrs, sizes, len(outs)))
        return outs

    def debug_read(self, what, level, frame=0):
        w, h, _, _ = self.level_dims(level, frame)
        written = C.c_size_t(0)
        if what == 0:
            out = np.zeros((h, w), np.uint8)
            check(self._L.ydorb_extractor_debug_read(self._h, 0, frame, level, _p(out), out.size, C.byref(written)))
            return out
        if what == 3:
            out = np.zeros(1, np.uint8)
            check(self._L.ydorb_extractor_debug_read(self._h, 3, frame, level, _p(out), 1, C.byref(written)))
            return int(out[0])
        out = np.zeros(1 << 17, KP_DTYPE)
        check(self._L.ydorb_extractor_debug_read(self._h, what, frame, level, _p(out), out.nbytes, C.byref(written)))
        return out[: written.value // KP_DTYPE.itemsize].copy()

    def set_profiling(self, on=True):
        check(self._L.ydorb_extractor_set_profiling(self._h, int(on)))

    def stage_times(self):
        names = (C.c_char_p * 16)()
        ms = (C.c_float * 16)()
        n = C.c_int32(0)
        check(self._L.ydorb_extractor_stage_times(self._h, 16, names, ms, C.byref(n)))
        return {names[i].decode(): ms[i] for i in range(n.value)}
