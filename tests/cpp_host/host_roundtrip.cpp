// TEST-ONLY: a plain C++ host (no Python, no OpenCV, no torch) on the C ABI of include/ydorb/c_api.h - what the reference's own
// translation units do once src/orbExtractor.cpp / orbMatcher.cpp / optimizer.cpp forward to the library.
//   host_roundtrip <in.raw> <w> <h> <out.bin>      (and: host_roundtrip ba <problem.bin> <out.bin>, below)
// reads an 8-bit gray image, extracts it (OrbExtractor(1000, 1.2, 8, 20, 7)), extracts a copy moved by (3, 2) px, matches the two frames
// with the searchByProjectionInLastAndCurrentFrame rules and writes keypoints, descriptors and the assignment to <out.bin>;
// tests/test_cpp_host.py compares that file with what the ctypes mirror returns for the same input.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ydorb/c_api.h"

#define CHECK(call)                                                              \
  do {                                                                           \
    const int rc_ = (call);                                                      \
    if (rc_ != YDORB_OK) {                                                       \
      std::fprintf(stderr, "%s -> %d: %s\n", #call, rc_, ydorb_last_error());    \
      return 1;                                                                  \
    }                                                                            \
  } while (0)

// host_roundtrip ba <problem.bin> <out.bin>: Optimizer::localBundleAdjust's flat problem (K, P, E, then poses K x 7 f64, fixed K u8,
// points P x 3 f64, edge_pose E i32, edge_point E i32, meas E x 3 f64, inv_sigma2 E f64, camera 5 f64) -> poses, points, outlier mask.
static int runBa(const char* in, const char* out) {
  FILE* f = std::fopen(in, "rb");
  if (!f) return 2;
  int32_t dims[3];
  if (std::fread(dims, sizeof(dims), 1, f) != 1) return 2;
  const int K = dims[0], P = dims[1], E = dims[2];
  std::vector<double> poses((size_t)K * 7), points((size_t)P * 3), meas((size_t)E * 3), info(E), cam(5);
  std::vector<uint8_t> fixed(K), outlier(E ? E : 1);
  std::vector<int32_t> ep(E), eq(E);
  bool ok = std::fread(poses.data(), 8, poses.size(), f) == poses.size() && std::fread(fixed.data(), 1, K, f) == (size_t)K &&
            std::fread(points.data(), 8, points.size(), f) == points.size() && std::fread(ep.data(), 4, E, f) == (size_t)E &&
            std::fread(eq.data(), 4, E, f) == (size_t)E && std::fread(meas.data(), 8, meas.size(), f) == meas.size() &&
            std::fread(info.data(), 8, E, f) == (size_t)E && std::fread(cam.data(), 8, 5, f) == 5;
  std::fclose(f);
  if (!ok) { std::fprintf(stderr, "short problem file\n"); return 2; }
  YdBaProblem prob{K, P, E, poses.data(), fixed.data(), points.data(), ep.data(), eq.data(), meas.data(), info.data(),
                   cam[0], cam[1], cam[2], cam[3], cam[4], nullptr};
  YdBaOptions opt;
  ydorb_ba_default_options(&opt);
  YdBaResult res;
  std::memset(&res, 0, sizeof(res));
  res.edge_outlier = outlier.data();
  CHECK(ydorb_ba_solve(&prob, &opt, &res));
  FILE* o = std::fopen(out, "wb");
  if (!o) return 2;
  const int32_t head[2] = {res.n_trials, res.n_iterations};
  std::fwrite(head, sizeof(head), 1, o);
  std::fwrite(poses.data(), 8, poses.size(), o);
  std::fwrite(points.data(), 8, points.size(), o);
  std::fwrite(outlier.data(), 1, E, o);
  std::fclose(o);
  std::printf("host_roundtrip ba: %d LM trials, chi2 %.6f\n", res.n_trials, res.n_log ? res.log_chi2[res.n_log - 1] : 0.0);
  return 0;
}

int main(int argc, char** argv) {
  if (argc == 4 && std::strcmp(argv[1], "ba") == 0) return runBa(argv[2], argv[3]);
  if (argc != 5) { std::fprintf(stderr, "usage: %s in.raw w h out.bin\n", argv[0]); return 2; }
  const int w = std::atoi(argv[2]), h = std::atoi(argv[3]);
  std::vector<uint8_t> img((size_t)w * h), moved((size_t)w * h);
  FILE* f = std::fopen(argv[1], "rb");
  if (!f || std::fread(img.data(), 1, img.size(), f) != img.size()) { std::fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
  std::fclose(f);
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) moved[(size_t)y * w + x] = img[(size_t)((y + h - 2) % h) * w + (x + w - 3) % w];

  YdExtractorConfig cfg{1000, 1.2f, 8, 20, 7, 0, 1, 0};
  ydorb_extractor_t* ex = nullptr;
  CHECK(ydorb_extractor_create(&cfg, &ex));
  const int cap = ydorb_extractor_max_keypoints(ex);
  std::vector<YdKeyPoint> ka(cap), kb(cap);
  std::vector<uint8_t> da((size_t)cap * 32), db((size_t)cap * 32);
  int32_t na = 0, nb = 0;
  CHECK(ydorb_extract(ex, img.data(), w, h, w, ka.data(), da.data(), cap, &na));
  CHECK(ydorb_extract(ex, moved.data(), w, h, w, kb.data(), db.data(), cap, &nb));
  float scale[8];
  CHECK(ydorb_extractor_tables(ex, scale, nullptr, nullptr, nullptr, nullptr));

  // queries = frame A's keypoints at their own position (OrbMatcher::searchByProjectionInLastAndCurrentFrame, orbMatcher.cpp:65-155)
  std::vector<YdQuery> q(na);
  for (int i = 0; i < na; i++) {
    YdQuery& Q = q[i];
    Q.u = ka[i].x; Q.v = ka[i].y; Q.r = 15.0f * scale[ka[i].octave];
    Q.min_level = ka[i].octave - 1; Q.max_level = ka[i].octave + 1;
    Q.ur = 0; Q.rs = 0; Q.angle = ka[i].angle; Q.level = ka[i].octave; Q.flags = 3;
  }
  YdFrameView fv{kb.data(), db.data(), nullptr, nb, 0.0f, (float)w, 0.0f, (float)h};
  ydorb_matcher_t* m = nullptr;
  CHECK(ydorb_matcher_create(0, &m));
  std::vector<int32_t> assigned(nb, -1);
  std::vector<uint8_t> taken(nb, 0);
  int32_t nMatches = 0;
  CHECK(ydorb_search_by_projection(m, YDORB_SEARCH_LAST_CURRENT, &fv, q.data(), da.data(), na, 0.9f, 0, 1, taken.data(), assigned.data(), &nMatches));

  FILE* o = std::fopen(argv[4], "wb");
  if (!o) return 2;
  const int32_t head[4] = {na, nb, nMatches, cap};
  std::fwrite(head, sizeof(head), 1, o);
  std::fwrite(ka.data(), sizeof(YdKeyPoint), na, o);
  std::fwrite(da.data(), 32, na, o);
  std::fwrite(kb.data(), sizeof(YdKeyPoint), nb, o);
  std::fwrite(db.data(), 32, nb, o);
  std::fwrite(assigned.data(), sizeof(int32_t), nb, o);
  std::fclose(o);
  ydorb_matcher_destroy(m);
  ydorb_extractor_destroy(ex);
  std::printf("host_roundtrip: %d + %d keypoints, %d matches\n", na, nb, nMatches);
  return 0;
}
