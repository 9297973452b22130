"""Python mirror of YDORBSLAM::Optimizer::localBundleAdjust (reference src/optimizer.hpp:34, optimizer.cpp:138-352) over the C ABI.

The graph is passed flat (what optimizer.cpp:175-283 hands to g2o); the covisibility walk and the map write-back belong to the caller.
"""
import ctypes as C
import numpy as np

from ._lib import BA_ALLREDUCE_FN, YdBaOptions, YdBaProblem, YdBaResult, YdPoseBatch, check, lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Optimizer:
    @staticmethod
    def default_options(device=0, phase_times=False):
        """phase_times: also fill the per-phase device times of the result (YDORB_BA_PHASE_TIMES: event pairs on the stream, ~8 % slower)."""
        from ._lib import BA_PHASE_TIMES
        o = YdBaOptions()
        lib().ydorb_ba_default_options(C.byref(o))
        o.device = device
        if phase_times:
            o.flags |= BA_PHASE_TIMES
        return o

    @staticmethod
    def global_options(iters=5, robust=True, device=0):
        """Options of Optimizer::bundleAdjust / globalBundleAdjust (optimizer.cpp:7-137, :353-357): one optimize(iters) call."""
        from ._lib import BA_NO_ROBUST, BA_SINGLE_STAGE
        o = Optimizer.default_options(device)
        o.iters1, o.iters2 = iters, 0
        o.delta_mono = float(np.float32(np.sqrt(5.99)))    # `const float monoDelta = sqrt(5.99)`, :37
        o.flags = BA_SINGLE_STAGE | (0 if robust else BA_NO_ROBUST)
        return o

    @staticmethod
    def local_bundle_adjust(prob, options=None, stop=None, allreduce=None, comm_tensor_ptr=None, comm_doubles=0, rank=0, world=1):
        """prob: dict(poses[K,7], fixed[K], points[P,3], edge_pose, edge_point, meas[E,3], info[E], camera[5]).
        Returns dict(poses, points, outlier, log (chi2, lambda, trials, stage), trials, iterations, ms)."""
        L = lib()
        o = options if options is not None else Optimizer.default_options()
        poses = np.ascontiguousarray(prob["poses"], np.float64).copy()
        points = np.ascontiguousarray(prob["points"], np.float64).copy()
        fixed = np.ascontiguousarray(prob["fixed"], np.uint8)
        ep = np.ascontiguousarray(prob["edge_pose"], np.int32)
        eq = np.ascontiguousarray(prob["edge_point"], np.int32)
        meas = np.ascontiguousarray(prob["meas"], np.float64)
        info = np.ascontiguousarray(prob["info"], np.float64)
        cam = [float(v) for v in prob["camera"]]
        E = len(ep)
        outlier = np.zeros(max(E, 1), np.uint8)
        P = YdBaProblem(len(poses), len(points), E, _p(poses), _p(fixed), _p(points), _p(ep), _p(eq), _p(meas), _p(info), *cam,
                        None if stop is None else _p(stop))
        cb = None
        if allreduce is not None:
            cb = BA_ALLREDUCE_FN(allreduce)
            o.allreduce = cb
            o.d_comm_buf = comm_tensor_ptr
            o.comm_doubles = comm_doubles
            o.rank, o.world = rank, world
        res = YdBaResult()
        res.edge_outlier = outlier.ctypes.data_as(C.c_void_p).value
        check(L.ydorb_ba_solve(C.byref(P), C.byref(o), C.byref(res)))
        n = res.n_log
        log = np.stack([np.array(res.log_chi2[:n]), np.array(res.log_lambda[:n]), np.array(res.log_trials[:n], np.float64),
                        np.array(res.log_stage[:n], np.float64)], axis=1) if n else np.zeros((0, 4))
        return dict(poses=poses, points=points, outlier=outlier[:E], log=log, trials=res.n_trials, iterations=res.n_iterations,
                    stopped=bool(res.stopped),
                    ms=dict(total=res.ms_total, errors=res.ms_errors, build=res.ms_build, schur=res.ms_schur, solve=res.ms_solve,
                            update=res.ms_update))

    @staticmethod
    def release(device=0):
        """ydorb_ba_release: give the solver's pooled device scratch on `device` back (the next solve allocates again)."""
        check(lib().ydorb_ba_release(device))

    @staticmethod
    def local_bundle_adjust_batch(probs, options=None, threads=0):
        """n independent problems in one ydorb_ba_solve_batch call (host threads inside the library, one pooled context each).
        Returns a list of the dicts local_bundle_adjust returns."""
        L = lib()
        o = options if options is not None else Optimizer.default_options()
        n = len(probs)
        if n == 0:
            return []
        P = (YdBaProblem * n)()
        R = (YdBaResult * n)()
        keep = []
        for i, prob in enumerate(probs):
            poses = np.ascontiguousarray(prob["poses"], np.float64).copy()
            points = np.ascontiguousarray(prob["points"], np.float64).copy()
            fixed = np.ascontiguousarray(prob["fixed"], np.uint8)
            ep = np.ascontiguousarray(prob["edge_pose"], np.int32)
            eq = np.ascontiguousarray(prob["edge_point"], np.int32)
            meas = np.ascontiguousarray(prob["meas"], np.float64)
            info = np.ascontiguousarray(prob["info"], np.float64)
            outlier = np.zeros(max(len(ep), 1), np.uint8)
            cam = [float(v) for v in prob["camera"]]
            P[i] = YdBaProblem(len(poses), len(points), len(ep), _p(poses), _p(fixed), _p(points), _p(ep), _p(eq), _p(meas), _p(info), *cam, None)
            R[i].edge_outlier = outlier.ctypes.data_as(C.c_void_p).value
            keep.append((poses, points, fixed, ep, eq, meas, info, outlier))
        check(L.ydorb_ba_solve_batch(P, n, C.byref(o), R, int(threads), None))
        out = []
        for i in range(n):
            res, (poses, points, _, ep, _, _, _, outlier) = R[i], keep[i]
            k = res.n_log
            log = np.stack([np.array(res.log_chi2[:k]), np.array(res.log_lambda[:k]), np.array(res.log_trials[:k], np.float64),
                            np.array(res.log_stage[:k], np.float64)], axis=1) if k else np.zeros((0, 4))
            out.append(dict(poses=poses, points=points, outlier=outlier[:len(ep)], log=log, trials=res.n_trials, iterations=res.n_iterations,
                            stopped=bool(res.stopped)))
        return out

    @staticmethod
    def optimize_poses(probs, device=0):
        """Optimizer::optimizePose (optimizer.cpp:358-501) for a batch of frames in one launch.  probs: list of dicts as
        ydorbslam_amd.synth.synth_pose_problem (pose[7], points[E,3], meas[E,3], info[E], camera[5] — the camera of probs[0] is used).
        Returns a list of dict(pose, outlier, inliers, chi2[4], trials)."""
        n = len(probs)
        if n == 0:
            return []
        counts = [len(p["info"]) for p in probs]
        start = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
        E = int(start[-1])
        poses = np.ascontiguousarray(np.stack([p["pose"] for p in probs]), np.float64).copy()
        cat = lambda k, w: (np.ascontiguousarray(np.concatenate([np.asarray(p[k], np.float64).reshape(-1, w) for p in probs]), np.float64)
                            if E else np.zeros((0, w)))
        X, z, wgt = cat("points", 3), cat("meas", 3), cat("info", 1)
        cam = [float(v) for v in probs[0]["camera"]]
        B = YdPoseBatch(n, device, _p(start), _p(poses), _p(X), _p(z), _p(wgt), *cam)
        outlier = np.zeros(max(E, 1), np.uint8)
        inl = np.zeros(n, np.int32); chi = np.zeros((n, 4), np.float64); trials = np.zeros(n, np.int32)
        check(lib().ydorb_pose_optimize(C.byref(B), _p(outlier), _p(inl), _p(chi), _p(trials)))
        return [dict(pose=poses[f].copy(), outlier=outlier[start[f]:start[f + 1]].copy(), inliers=int(inl[f]), chi2=chi[f].copy(),
                     trials=int(trials[f])) for f in range(n)]

    @staticmethod
    def dense_solve(A, b, device=0):
        A = np.ascontiguousarray(A, np.float64)
        b = np.ascontiguousarray(b, np.float64)
        x = np.zeros(len(b), np.float64)
        ok = C.c_int32(0)
        check(lib().ydorb_ba_dense_solve(device, _p(A), len(b), _p(b), _p(x), C.byref(ok)))
        return x, bool(ok.value)
