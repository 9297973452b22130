/* ydorb C ABI — the drop-in boundary of the MI355X-native ORB / matcher / local-BA hot path.
 *
 * The reference (WeiZhang1988/YDORBSLAM) has no FFI layer: its hot path sits behind three C++
 * classes linked into libslam.so (src/CMakeLists.txt:14-31).  This header is the flat C ABI a
 * maintainer binds those classes to; the .hpp files next to this header are the adapter classes with the
 * reference's own signatures.  Every entry point cites the reference interface it replaces.
 *
 * Conventions: plain pointers + sizes, caller-owned buffers, opaque handles, int status return
 * (0 = YDORB_OK, <0 = error; ydorb_last_error() gives text).  No exceptions cross the ABI.
 * "d_" pointers are device (HBM) addresses, everything else is host memory.
 * A handle owns one HIP stream; distinct handles may be used concurrently from distinct threads
 * (the reference runs two extractors on two transient threads, src/frame.cpp:84-87).
 */
#ifndef YDORB_C_API_H
#define YDORB_C_API_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define YDORB_OK 0
#define YDORB_ERR_INVALID_ARG -1
#define YDORB_ERR_NO_DEVICE -2     /* HIP runtime / gfx950 device not usable: there is NO CPU fallback */
#define YDORB_ERR_HIP -3           /* a HIP call failed */
#define YDORB_ERR_CAPACITY -4      /* caller buffer or internal scratch too small */
#define YDORB_ERR_UNSUPPORTED -5
#define YDORB_ERR_NUMERIC -6       /* BA: non-finite state */

const char* ydorb_last_error(void);
/* number of usable gfx950 devices (0 if none); never throws */
int ydorb_device_count(void);
const char* ydorb_version(void);

/* ------------------------------------------------------------------------------------------
 * cv::KeyPoint-compatible POD: 7 x 4 bytes, no padding (what extractAndCompute fills,
 * src/orbExtractor.cpp:355-399).
 * ---------------------------------------------------------------------------------------- */
typedef struct YdKeyPoint {
  float x, y;       /* pt, level-0 pixel units (level coords * scaleFactor[octave]) */
  float size;       /* (float)(int)(31 * scaleFactor[octave])   orbExtractor.cpp:595,600 */
  float angle;      /* degrees [0,360), cv::fastAtan2            orbExtractor.cpp:419 */
  float response;   /* FAST-9/16 corner score */
  int32_t octave;
  int32_t class_id; /* -1 */
} YdKeyPoint;

/* ------------------------------------------------------------------------------------------
 * ORB extractor.  Replaces YDORBSLAM::OrbExtractor (src/orbExtractor.hpp:31-74).
 * ---------------------------------------------------------------------------------------- */
typedef struct YdExtractorConfig {
  int32_t n_features;    /* OrbExtractor ctor args, src/orbExtractor.cpp:315-317 */
  float scale_factor;
  int32_t n_levels;      /* 1..8 */
  int32_t ini_fast_thr;
  int32_t min_fast_thr;  /* accepted and, like the reference (:318), replaced by ini_fast_thr */
  int32_t device;        /* HIP device ordinal */
  int32_t max_batch;     /* frames per launch the scratch is sized for (>=1) */
  int32_t flags;         /* YDORB_EXTRACTOR_* (0 = defaults) */
} YdExtractorConfig;
/* Every launch of a call goes on the one stream the call runs on (the quad-tree launches otherwise use two side streams of the handle
 * and overlap the blur: ~5 % faster alone).  For callers that pipeline several handles on several streams: the device has 4 hardware
 * queues, streams share them round-robin, and a side stream that lands behind another stream's long kernel stalls its handle. */
#define YDORB_EXTRACTOR_SINGLE_STREAM 1

typedef struct ydorb_extractor ydorb_extractor_t;

/* OrbExtractor::OrbExtractor, src/orbExtractor.cpp:315-354 */
int ydorb_extractor_create(const YdExtractorConfig* cfg, ydorb_extractor_t** out);
void ydorb_extractor_destroy(ydorb_extractor_t* h);

/* getScaleFactors / getInvScaleFactors / getScaleFactorSquares / getInvScaleFactorSquares and the
 * per-level keypoint quotas (src/orbExtractor.hpp:42-49, orbExtractor.cpp:325-339).  Each output may
 * be NULL; arrays hold n_levels entries. */
int ydorb_extractor_tables(const ydorb_extractor_t* h, float* scale, float* inv_scale, float* scale_sq,
                           float* inv_scale_sq, int32_t* per_level);
/* upper bound of keypoints one frame can return (sum of the per-level quotas) */
int ydorb_extractor_max_keypoints(const ydorb_extractor_t* h);

/* OrbExtractor::extractAndCompute(image, keypoints, descriptors), src/orbExtractor.cpp:355-399.
 * img: 8-bit gray, `stride` bytes per row (host).  kps/desc: caller buffers for `cap` keypoints
 * (desc is cap x 32 bytes, row-major like the CV_8U Mat at :373).  *n_out = keypoints written.
 * Empty image (w<=0||h<=0||!img) returns YDORB_OK with *n_out = 0, like :357-359. */
int ydorb_extract(ydorb_extractor_t* h, const uint8_t* img, int32_t w, int32_t hgt, int32_t stride,
                  YdKeyPoint* kps, uint8_t* desc, int32_t cap, int32_t* n_out);

/* Batched form of the same call: n_frames images of identical size, frame f at img + f*frame_stride.
 * Outputs: frame f's keypoints at kps + f*cap, descriptors at desc + f*cap*32, count at n_out[f]. */
int ydorb_extract_batch(ydorb_extractor_t* h, const uint8_t* img, int32_t w, int32_t hgt, int32_t stride,
                        size_t frame_stride, int32_t n_frames, YdKeyPoint* kps, uint8_t* desc, int32_t cap,
                        int32_t* n_out);

/* Device-resident batched form (inputs and outputs in HBM, asynchronous on the handle's stream or
 * on `stream` (a hipStream_t) when non-NULL; no host synchronisation).  cap must be >=
 * ydorb_extractor_max_keypoints().  d_n_out: int32[n_frames]. */
int ydorb_extract_batch_device(ydorb_extractor_t* h, const uint8_t* d_img, int32_t w, int32_t hgt, int32_t stride,
                               size_t frame_stride, int32_t n_frames, YdKeyPoint* d_kps, uint8_t* d_desc,
                               int32_t cap, int32_t* d_n_out, void* stream);
int ydorb_extractor_synchronize(ydorb_extractor_t* h);

/* m_v_imagePyramid[level] of the last call (public member read by src/frame.cpp:366,412-427):
 * device pointer to the level's ROI origin inside its 19-px reflect-101 padded buffer. */
int ydorb_extractor_pyramid(const ydorb_extractor_t* h, int32_t frame, int32_t level, const uint8_t** d_roi,
                            int32_t* w, int32_t* hgt, int32_t* stride);
/* Copy one padded level ((hgt+38) rows x (w+38) bytes, tightly packed) of the last call to host. */
int ydorb_extractor_read_level(ydorb_extractor_t* h, int32_t frame, int32_t level, uint8_t* dst, size_t dst_bytes);
/* All levels 0 .. n_levels-1 of one frame in ONE device-to-host transfer: dst_levels[l] receives (hgt_l+38) rows x (w_l+38) bytes,
 * tightly packed (the layout of ydorb_extractor_read_level).  This is what the adapter's m_v_imagePyramid refresh uses. */
int ydorb_extractor_read_pyramid(ydorb_extractor_t* h, int32_t frame, uint8_t* const* dst_levels, const size_t* dst_bytes, int32_t n_levels);

/* Test/diagnostic access to intermediate stages of the last call (parity tests compare each stage
 * with the oracle).  what: 0 = blurred level (hgt x w bytes), 1 = pre-quad-tree candidates
 * (YdKeyPoint[], border-relative coords as at orbExtractor.cpp:587-589), 2 = per-level keypoints after
 * orientation (YdKeyPoint[], level coords), 3 = one byte: which quad-tree kernel produced the (frame, level) unit
 * (0 = flat histogram/sort form, 1 = pass form; YDORB_QT_PASS=1 in the environment at create forces 1).
 * Returns bytes written in *written. */
int ydorb_extractor_debug_read(ydorb_extractor_t* h, int32_t what, int32_t frame, int32_t level, void* dst,
                               size_t dst_bytes, size_t* written);

/* Average device time (ms) of each pipeline stage over the calls since the last reset, measured with
 * HIP events on the handle's stream when profiling is enabled.  names: out array of static strings. */
int ydorb_extractor_set_profiling(ydorb_extractor_t* h, int32_t on);
int ydorb_extractor_stage_times(ydorb_extractor_t* h, int32_t max_stages, const char** names, float* ms,
                                int32_t* n_stages);

/* ------------------------------------------------------------------------------------------
 * Descriptor matcher.  Replaces YDORBSLAM::OrbMatcher (src/orbMatcher.hpp:24-66) for the
 * search-by-projection and search-by-BoW families.  Frame / KeyFrame / MapPoint objects do not cross the
 * ABI: a map point is a query row (descriptor + projected position + flags) and `assigned[idx]` stands
 * for frame.m_v_sptrMapPoints[idx] (a query index, or -1 for null).  The float geometry that projects map
 * points (cv::Mat products at orbMatcher.cpp:84-93,171-177; Frame::isInCameraFrustum, frame.cpp:295-326)
 * stays in the adapter class, which fills YdQuery with the very floats the reference computes.
 * ---------------------------------------------------------------------------------------- */
typedef struct YdQuery {
  float u, v;          /* projected image position */
  float r;             /* radius given to Frame::getKeyPointsInArea (frame.cpp:337) */
  int32_t min_level, max_level; /* its _minScaleLevel/_maxScaleLevel arguments (-1 = open) */
  float ur;            /* projected right-image x; stereo test |ur - rightX[idx]| <= rs (orbMatcher.cpp:41,109) */
  float rs;
  float angle;         /* angle of the source keypoint (rotation histogram, orbMatcher.cpp:121) */
  int32_t level;       /* predicted / source scale level */
  int32_t flags;       /* bit0: query takes part; bit1: its map point has observations > 0 */
} YdQuery;

typedef struct YdFrameView {   /* the parts of YDORBSLAM::Frame the searches read (src/frame.hpp) */
  const YdKeyPoint* kps;       /* m_v_keyPoints (undistorted) */
  const uint8_t* desc;         /* m_cvMat_descriptors, n x 32 */
  const float* right_x;        /* m_v_rightXcords or NULL */
  int32_t n;
  float min_x, max_x, min_y, max_y; /* m_flt_minX.. (image bounds; the 64x48 grid spans them, frame.cpp:99-100) */
} YdFrameView;

typedef struct ydorb_matcher ydorb_matcher_t;
int ydorb_matcher_create(int32_t device, ydorb_matcher_t** out);
void ydorb_matcher_destroy(ydorb_matcher_t* h);

/* static int OrbMatcher::computeDescriptorsDistance(a, b), src/orbMatcher.cpp:11-23 (one pair, host). */
int ydorb_descriptor_distance(const uint8_t* a32, const uint8_t* b32);
/* The same distance for n row pairs on the GPU: out[i] = d(a[i], b[i]) (host pointers). */
int ydorb_descriptor_distance_rows(ydorb_matcher_t* h, const uint8_t* a, const uint8_t* b, int32_t n, int32_t* out);

/* Frame::getKeyPointsInArea, src/frame.cpp:337-361, on the GPU grid (ordered as the reference returns them). */
int ydorb_frame_keypoints_in_area(ydorb_matcher_t* h, const YdFrameView* frame, float x, float y, float r, int32_t min_level,
                                  int32_t max_level, int32_t* out_idx, int32_t cap, int32_t* n_out);

#define YDORB_SEARCH_FRAME_MAPPOINT 0   /* searchByProjectionInFrameAndMapPoint,       orbMatcher.cpp:24-64   */
#define YDORB_SEARCH_LAST_CURRENT 1     /* searchByProjectionInLastAndCurrentFrame,    orbMatcher.cpp:65-155  */
#define YDORB_SEARCH_KEYFRAME_CURRENT 2 /* searchByProjectionInKeyFrameAndCurrentFrame, orbMatcher.cpp:156-239 */
#define YDORB_SEARCH_SIM_PROJECTION 7   /* searchByProjectionInSim,                    orbMatcher.cpp:240-302 */
#define YDORB_SEARCH_BOW_KEYFRAME_FRAME 3 /* searchByBowInKeyFrameAndFrame,            orbMatcher.cpp:303-379 */
#define YDORB_SEARCH_BOW_TWO_KEYFRAMES 4  /* searchByBowInTwoKeyFrames,                orbMatcher.cpp:380-462 */

/* Projection family.  taken[idx] (in/out, n bytes): 1 where the frame keypoint already holds a map point that
 * blocks it (observations > 0 for modes 0/1, any map point for mode 2).  assigned[idx] (in/out, n ints): query
 * index written where the reference writes frame.m_v_sptrMapPoints[bestIdx]; untouched entries keep their value.
 * *n_matches = the int the reference returns (it counts overwrites and subtracts histogram culls exactly as
 * orbMatcher.cpp does).  * mode 7 = OrbMatcher::searchByProjectionInSim (orbMatcher.cpp:240-302, loop closing): the caller projects the map points with the
 * Sim3 it decomposed as the reference does; window without level check, explicit level window query.level-1 .. query.level, taken[idx] =
 * "_vSptrMatchedMapPoints[idx] is set" (a match takes its feature), best distance <= 50, no orientation check. */
int ydorb_search_by_projection(ydorb_matcher_t* h, int32_t mode, const YdFrameView* frame, const YdQuery* queries,
                               const uint8_t* qdesc, int32_t nq, float ratio, int32_t orb_dist, int32_t check_orientation,
                               uint8_t* taken, int32_t* assigned, int32_t* n_matches);

/* DBoW3::FeatureVector (std::map<node, vector<feature>>) as CSR: node_ids ascending, node_start[n_nodes+1], feat[]. */
typedef struct YdFeatureVector {
  const uint32_t* node_ids;
  const int32_t* node_start;
  const int32_t* feat;
  int32_t n_nodes;
} YdFeatureVector;
typedef struct YdBowSide {
  const YdKeyPoint* kps;
  const uint8_t* desc;
  const uint8_t* valid;   /* 1 where the feature has a good map point (may be NULL for the frame side of mode 3) */
  int32_t n;
  YdFeatureVector fv;
} YdBowSide;
/* BoW family.  mode 3: out[n_b] = keyframe-A feature whose map point goes to frame feature idx (or -1);
 * mode 4: out[n_a] = matched feature of keyframe B (or -1). */
int ydorb_search_by_bow(ydorb_matcher_t* h, int32_t mode, const YdBowSide* a, const YdBowSide* b, float ratio,
                        int32_t check_orientation, int32_t* out, int32_t* n_matches);


/* Search half of OrbMatcher::fuseByProjection (src/orbMatcher.cpp:682-745; fuseBySim3 :746-807 runs the same test on a
 * Sim3-projected point).  One query per map point that passed the reference's own predicates (not bad, not already in the keyframe,
 * in front of the camera, inside the image, inside the distance invariance, viewing angle — flags bit 0 set; other queries are
 * skipped): u, v = projection, r = th * scaleFactor[predicted level], ur = projected right x, level = predicted level.
 * best_idx[q] = the keyframe feature with the smallest descriptor distance (<= 50) among the window's features at level
 * predicted-1 .. predicted whose squared reprojection error passes the chi-square test (5.99 mono / 7.81 stereo), or -1.
 * The caller then walks the list in order and applies beReplacedBy / addObservation exactly as :726-737 (re-checking isBad /
 * isInKeyFrame at each step, because earlier replacements can change them; they cannot change a later point's search result). */
int ydorb_fuse_search(ydorb_matcher_t* h, const YdFrameView* keyframe, const YdQuery* queries, const uint8_t* qdesc, int32_t n_queries,
                      const float* inv_scale_factor_squares, int32_t n_levels, int32_t* best_idx, int32_t* n_found);
/* The same search with the acceptance distance as a parameter (ydorb_fuse_search = max_dist 50).  An all-zero inverse-sigma table
 * disables the chi-square test: that is fuseBySim3's search (orbMatcher.cpp:746-807) and, with max_dist = 100 (TH_HIGH), each of the two
 * directions of searchBySim3 (:594-667), whose agreement check (:668-679) the caller runs on the two result vectors. */
int ydorb_window_search(ydorb_matcher_t* h, const YdFrameView* keyframe, const YdQuery* queries, const uint8_t* qdesc, int32_t n_queries,
                        const float* inv_scale_factor_squares, int32_t n_levels, int32_t max_dist, int32_t* best_idx, int32_t* n_found);

/* OrbMatcher::searchForTriangulation (src/orbMatcher.cpp:463-565): BoW-guided search between the features of two
 * keyframes that have no MapPoint yet, kept when the second feature lies on the first one's epipolar line (:808-819).
 * has_map_point[i] != 0 <=> KeyFrame::getMapPoint(i) is set; right_x = m_v_rightXcords.  F = _fMatrix_first2second row-major
 * (F[r*3+c] = at<float>(r,c)); (epipole_x, epipole_y) = the first camera centre projected into the second image (:465-470, computed
 * by the caller exactly as the reference does); second_scale_factors / _squares = the second keyframe's m_v_scaleFactors /
 * m_v_scaleFactorSquares.  matched_second[first idx] = second idx or -1 (the reference's vector of pairs, in first-index order). */
typedef struct YdTriSide {
  const YdKeyPoint* kps;
  const uint8_t* desc;
  const float* right_x;
  const uint8_t* has_map_point;
  int32_t n;
  YdFeatureVector fv;
} YdTriSide;
int ydorb_search_for_triangulation(ydorb_matcher_t* h, const YdTriSide* first, const YdTriSide* second, const float* F, float epipole_x,
                                   float epipole_y, const float* second_scale_factors, const float* second_scale_factor_squares,
                                   int32_t n_levels, int32_t stereo_only, int32_t check_orientation, int32_t* matched_second,
                                   int32_t* n_matches);

/* MapPoint::computeDistinctiveDescriptors(), src/mapPoint.cpp:169-218, for a batch of map points.  Point p's descriptors (the
 * rows the member function collects at :183-187, in that order) are desc[offsets[p] .. offsets[p+1]); offsets[0] = 0.
 * best[p] = bestMedianIdx (:203-213): the first descriptor whose sorted distance row has the least entry at (int)(0.5 m);
 * -1 for a point without descriptors (the reference returns early, :188).  One map point per call is far below launch
 * latency - hand over all points of a LocalMapping / LoopClosing pass at once. */
int ydorb_distinctive_descriptors(ydorb_matcher_t* h, const uint8_t* desc, const int32_t* offsets, int32_t n_points, int32_t* best);

/* Frame::computeStereoMatches(), src/frame.cpp:362-477: for every left keypoint the best right descriptor among the right
 * keypoints whose row band covers its row, an 11x11 block match over 11 column shifts at the keypoint's pyramid level,
 * parabola refinement, disparity -> depth, and the final outlier rule.  One call handles n_pairs rectified pairs.
 * The image pyramids are those the two extractors built in their last call (m_v_imagePyramid, read at :366,412-427):
 * pair p reads frame first_frame + p*frame_step of that call on each side (so one batched extractor call over interleaved
 * left/right images, or two extractors, both work).  Keypoints are the undistorted ones (frame.cpp:91-93).
 * right_x / depth: float [n_pairs][left->cap] = m_v_rightXcords / m_v_depth (-1 = none, -2 = removed by the outlier rule).
 * n_kept[p] = entries of vDistIndices; status[p]: bit0 = a left keypoint's (int)pt.y was outside [0, rows) (undefined vector
 * index in the reference; counted as a row without right keypoints), bit1 = a block-match window started left of the image
 * (cv::Exception in the reference; treated like the other window `continue`s).  Either may be NULL.
 * flags: default 0 replays the reference as written, including its left index that only advances when a keypoint reaches
 * the end of the loop body (:462) - a serial chain over the keypoints of a pair; YDORB_STEREO_INDEX_BY_KEYPOINT uses the
 * keypoint's own index for the descriptor row and the output slot (one wave per keypoint).  With
 * YDORB_STEREO_DEVICE_POINTERS every array argument is a device pointer and the call is asynchronous on `stream`
 * (a hipStream_t) or the matcher's own, ordered by the caller after the extractor calls. */
#define YDORB_STEREO_INDEX_BY_KEYPOINT 1
#define YDORB_STEREO_DEVICE_POINTERS 2
typedef struct YdStereoSide {
  const ydorb_extractor_t* extractor;
  int32_t first_frame, frame_step;
  const YdKeyPoint* kps;  /* [n_pairs][cap] */
  const uint8_t* desc;    /* [n_pairs][cap][32] */
  const int32_t* n;       /* [n_pairs] */
  int32_t cap;            /* right side: <= 8192 */
  int32_t reserved;
} YdStereoSide;
int ydorb_stereo_matches(ydorb_matcher_t* h, const YdStereoSide* left, const YdStereoSide* right, int32_t n_pairs, float bf, float b,
                         int32_t flags, float* right_x, float* depth, int32_t* n_kept, int32_t* status, void* stream);

/* Device-resident streaming form used after ydorb_extract_batch_device: for f = 0..n_frames-2, the keypoints of
 * frame f are searched in frame f+1 with the searchByProjectionInLastAndCurrentFrame rules (mode 1; position
 * prediction = d_affine[f] (2x3, row-major) applied to the keypoint, NULL = identity; window th*scaleFactor[octave];
 * levels octave-1..octave+1).  d_assigned: int32 [n_frames-1][cap] (query index per frame-f+1 keypoint or -1),
 * d_counts: int32 [n_frames-1].  Asynchronous on `stream` (or the matcher's own). */
int ydorb_match_consecutive_device(ydorb_matcher_t* h, const YdKeyPoint* d_kps, const uint8_t* d_desc, const int32_t* d_n,
                                   int32_t cap, int32_t n_frames, int32_t width, int32_t height, float th,
                                   const float* scale_factors, int32_t n_levels, const float* d_affine, int32_t check_orientation,
                                   int32_t* d_assigned, int32_t* d_counts, void* stream);
/* The same search for an explicit list of (query frame, target frame) pairs over two device-resident frame sets (which may be the
 * same set): pair p searches the keypoints of queries[pairs[2p]] in targets[pairs[2p+1]].  This is the form the multi-GPU path
 * uses after the all-gather of every rank's [keypoints | descriptors | count] records: targets = the gathered set of all ranks,
 * queries = the frames whose successor this rank owns; greedy acceptance stays with the owner of the target frame.
 * `pairs` is a host array; both sets must use the same cap.  d_assigned: int32 [n_pairs][cap] indexed by target keypoint,
 * d_counts: int32 [n_pairs], d_affine: [n_pairs][6] or NULL.  Asynchronous on `stream` (or the matcher's own). */
typedef struct YdFrameSetDev {
  const YdKeyPoint* d_kps;  /* [n_frames][cap] */
  const uint8_t* d_desc;    /* [n_frames][cap][32] */
  const int32_t* d_n;       /* [n_frames] */
  int32_t n_frames, cap;
} YdFrameSetDev;
int ydorb_match_pairs_device(ydorb_matcher_t* h, const YdFrameSetDev* queries, const YdFrameSetDev* targets, const int32_t* pairs,
                             int32_t n_pairs, int32_t width, int32_t height, float th, const float* scale_factors, int32_t n_levels,
                             const float* d_affine, int32_t check_orientation, int32_t* d_assigned, int32_t* d_counts, void* stream);
/* Brute-force 256-bit Hamming top-2 (BASELINE north_star).  For every query descriptor the result of the reference's
 * best / second-best chain (orbMatcher.cpp:39-52 and the same chain at :327-334, :404-417, :518-528) over its candidates in list
 * order: best = the FIRST candidate with the least distance, second = the first one with the least distance among the others;
 * both start at 256 and only a strictly smaller distance replaces them (dist 256 / index -1 = none).  *_rank = position in the
 * query's candidate list (what breaks ties), *_idx = the target row.
 * Host form: q [nq][32], t [nt][32]; candidates of query i = cand_idx[cand_offsets[i] .. cand_offsets[i+1]) (CSR), or - with
 * cand_offsets == cand_idx == NULL - all nt targets in index order.  nt <= 65535 per list. */
typedef struct YdMatch2 {
  int32_t best_dist, best_idx, second_dist, second_idx, best_rank, second_rank;
} YdMatch2;
int ydorb_hamming_topk(ydorb_matcher_t* h, const uint8_t* q, int32_t nq, const uint8_t* t, int32_t nt, const int32_t* cand_offsets,
                       const int32_t* cand_idx, YdMatch2* out);
/* Device-resident all-pairs form for n_pairs (query frame, target frame) pairs laid out [n_pairs][cap][32] with per-pair counts:
 * d_out [n_pairs][cap] YdMatch2.  Asynchronous on `stream` (or the matcher's own). */
int ydorb_hamming_topk_device(ydorb_matcher_t* h, const uint8_t* d_qdesc, const int32_t* d_nq, const uint8_t* d_tdesc, const int32_t* d_nt,
                              int32_t cap, int32_t n_pairs, YdMatch2* d_out, void* stream);
int ydorb_matcher_synchronize(ydorb_matcher_t* h);
/* average device ms of grid build / gather / resolve over calls since enabling (HIP events on the launch stream) */
int ydorb_matcher_set_profiling(ydorb_matcher_t* h, int32_t on);
int ydorb_matcher_stage_times(ydorb_matcher_t* h, int32_t max_stages, const char** names, float* ms, int32_t* n_stages);

/* ------------------------------------------------------------------------------------------
 * Vocabulary.  Replaces DBoW3::Vocabulary::transform(features, BowVector&, FeatureVector&, levelsup)
 * (thirdParty/DBow3/src/Vocabulary.cpp:752-824; descent :836-874) as called by Frame::computeBoW / KeyFrame::computeBoW
 * (src/frame.cpp:265-272, levelsup = 4).  The tree crosses the boundary once, as flat arrays (the adapter flattens
 * Vocabulary::m_nodes): node 0 is the root, node i's children are child_ids[child_begin[i] .. child_begin[i+1]) in the order of
 * Node::children (that order breaks distance ties: first minimum, :858-865); a node without children is a word.
 * ---------------------------------------------------------------------------------------- */
typedef struct YdVocabularyTree {
  int32_t n_nodes;
  int32_t levels;              /* m_L */
  const int32_t* child_begin;  /* [n_nodes + 1] */
  const int32_t* child_ids;    /* [child_begin[n_nodes]] */
  const uint8_t* node_desc;    /* [n_nodes][32]  Node::descriptor (the root's row is not read) */
  const double* node_weight;   /* [n_nodes]      Node::weight (read for words) */
  const int32_t* node_word;    /* [n_nodes]      Node::word_id (read for words) */
  int32_t weighting;           /* WeightingType: 0 TF_IDF, 1 TF, 2 IDF, 3 BINARY */
  int32_t norm;                /* what the scoring object's mustNormalize() asks for: 0 none, 1 L1, 2 L2 */
} YdVocabularyTree;
typedef struct ydorb_vocabulary ydorb_vocabulary_t;
int ydorb_vocabulary_create(const YdVocabularyTree* tree, int32_t device, ydorb_vocabulary_t** out);
void ydorb_vocabulary_destroy(ydorb_vocabulary_t* h);
/* One call transforms n_frames descriptor sets (frame f: n[f] <= cap rows at desc + f*cap*32; cap <= 8192).
 * BowVector of frame f: bow_word / bow_value [f*cap .. + n_words[f]) in ascending word id (std::map order), values exactly as the
 * reference's sequence of additions and its normalisation produce them (doubles).  FeatureVector of frame f as the CSR that
 * YdFeatureVector takes: fv_node [f*cap .. + n_fv_nodes[f]) ascending, fv_start [f*(cap+1) ..] offsets into fv_feat [f*cap ..].
 * status[f] (may be NULL) bit0: a descent ended above level L - levelsup, where the reference reads its `nid` uninitialised
 * (Vocabulary.cpp:777,868); such a feature is filed under the leaf's own node id.  Calls on one handle from several threads are
 * serialised inside (the reference shares one vocabulary between its tracking, local-mapping and loop-closing threads). */
int ydorb_vocabulary_transform(ydorb_vocabulary_t* h, const uint8_t* desc, const int32_t* n, int32_t n_frames, int32_t cap, int32_t levelsup,
                               int32_t* bow_word, double* bow_value, int32_t* n_words, int32_t* fv_node, int32_t* fv_start, int32_t* fv_feat,
                               int32_t* n_fv_nodes, int32_t* status);

/* ------------------------------------------------------------------------------------------
 * Local bundle adjustment.  Replaces the g2o work inside
 *   static void Optimizer::localBundleAdjust(shared_ptr<KeyFrame>, shared_ptr<Map>, bool& stop)   src/optimizer.cpp:138-352
 * (and the shared BA kernel of Optimizer::bundleAdjust, :7-137).  The covisibility walk that collects local /
 * fixed keyframes and map points (:140-173) and the write-back under Map::m_mutex_updateMap (:336-351) stay in the
 * adapter; this call receives the flat graph that :175-283 would hand to g2o and runs :284-334 on the GPU:
 * optimize(5) with Huber kernels -> mark edges with chi2 > 5.991 (mono) / 7.815 (stereo) or non-positive depth and drop
 * the kernels -> optimize(10) on the inliers -> final outlier list.
 * ---------------------------------------------------------------------------------------- */
typedef struct YdBaProblem {
  int32_t n_poses, n_points, n_edges;
  double* poses;                 /* [n_poses][7] tx,ty,tz,qx,qy,qz,qw = g2o::SE3Quat of T_c2w (Converter, converter.cpp:12-19); in/out */
  const uint8_t* pose_fixed;     /* [n_poses] setFixed(): keyframe id 0 and the "fixed keyframes" (optimizer.cpp:191,204) */
  double* points;                /* [n_points][3] world position; in/out */
  const int32_t* edge_pose;      /* [n_edges] index into poses */
  const int32_t* edge_point;     /* [n_edges] index into points */
  const double* edge_meas;       /* [n_edges][3] u, v, u_right; u_right < 0 selects the monocular 2-D edge (optimizer.cpp:239) */
  const double* edge_inv_sigma2; /* [n_edges] m_v_invScaleFactorSquares[octave] (information = I * this, :248,268) */
  double fx, fy, cx, cy, bf;     /* Frame::m_flt_* statics (:254-257, :275-279) */
  const volatile uint8_t* stop;  /* the reference's `bool& _bIsStopping`; NULL = the null-reference default (optimizer.hpp:34) */
} YdBaProblem;

typedef struct YdBaOptions {
  int32_t iters1, iters2;        /* 5 and 10 (optimizer.cpp:288,314) */
  double chi2_mono, chi2_stereo; /* 5.991, 7.815 (:296,306) */
  double delta_mono, delta_stereo; /* Huber delta: (double)(float)sqrt(5.991|7.815) (:223-224) */
  int32_t max_trials;            /* maxTrialsAfterFailure = 10 (optimization_algorithm_levenberg.cpp:50) */
  int32_t device;
  /* multi-GPU (landmarks sharded over ranks, every rank holds all poses): called on device buffer d_buf holding
   * `count` doubles; op 0 = sum, 1 = max over ranks.  NULL = single GPU. */
  int32_t (*allreduce)(void* user, void* d_buf, int64_t count, int32_t op);
  void* allreduce_user;
  void* d_comm_buf;              /* device workspace the callback can address (e.g. a torch tensor), >= comm_doubles doubles */
  int64_t comm_doubles;
  int32_t rank, world;
  int32_t flags;                 /* YDORB_BA_* bits below; 0 = localBundleAdjust's two-stage schedule */
  int32_t reserved;
} YdBaOptions;
/* Optimizer::bundleAdjust / globalBundleAdjust (optimizer.cpp:7-137, :353-357) use the same graph and solver with ONE
 * optimize(iters1) call: no chi2 cull, no second stage (edge_outlier is still filled with the chi2/depth test for the caller's
 * information), Huber kernels optional (_bIsRobust) and delta_mono = (double)(float)sqrt(5.99) — note 5.99, :37. */
#define YDORB_BA_SINGLE_STAGE 1
#define YDORB_BA_NO_ROBUST 2
/* fill YdBaResult.ms_errors .. ms_update: a pair of stream events round every phase of every LM trial, ~10 us of stream time each
 * (8 % of a 100-keyframe solve), so the breakdown is opt-in; ms_total is always measured */
#define YDORB_BA_PHASE_TIMES 4

typedef struct YdBaResult {
  int32_t n_trials;              /* LM trials executed (each = linearise/Schur/solve/update/chi2) */
  int32_t n_iterations;          /* outer LM iterations over both stages */
  int32_t n_log;
  int32_t stopped;               /* 1 if the stop flag ended the run early */
  double log_chi2[32];           /* robust chi2 after each outer iteration */
  double log_lambda[32];
  int32_t log_trials[32];
  int32_t log_stage[32];
  uint8_t* edge_outlier;         /* [n_edges] caller buffer: 1 = erase the observation (optimizer.cpp:316-334); may be NULL */
  float ms_total, ms_errors, ms_build, ms_schur, ms_solve, ms_update; /* device time per phase, summed over trials (HIP events) */
} YdBaResult;

/* default options as the reference uses them */
void ydorb_ba_default_options(YdBaOptions* opt);
/* Re-entrant: every call takes one of 8 per-device contexts (own stream and scratch), so several host threads — several maps or
 * sessions — may solve at once; a ninth concurrent caller waits.  Results do not depend on what else runs (fixed summation
 * orders).  The reference itself calls localBundleAdjust from one thread (localMapping.cpp:29). */
int ydorb_ba_solve(const YdBaProblem* prob, const YdBaOptions* opt, YdBaResult* res);
/* n independent problems (several maps / sessions / replayed windows) solved in LOCK STEP: one set of kernel launches per phase for
 * all problems (blockIdx.z = problem), the LM state of every problem kept on the host exactly as in ydorb_ba_solve and decided once per
 * round from ONE read-back.  One solve is a latency chain of ~35 small launches per LM trial that leaves most of the GPU idle; a
 * batch costs about the time of its slowest member plus the throughput-bound kernels.  `threads` = problems advanced together
 * (0 = up to 64 per group; larger batches run group after group).  The problems may differ in every size.  res[i] / rc_each[i]
 * (may be NULL) per problem; returns the first non-zero status.  Every result is bit-identical to its own ydorb_ba_solve call. */
int ydorb_ba_solve_batch(const YdBaProblem* probs, int32_t n, const YdBaOptions* opt, YdBaResult* res, int32_t threads, int32_t* rc_each);

/* The solver keeps its device scratch between calls: 8 contexts per device for ydorb_ba_solve / ydorb_pose_optimize and up to 64
 * problem contexts (~40 MB each at 100 keyframes x 10 000 points) for ydorb_ba_solve_batch.  ydorb_ba_release frees all of it on `device`
 * (waiting for solves in flight); the next call allocates again.  The reference has no counterpart: g2o's optimizer dies with
 * localBundleAdjust's stack frame (optimizer.cpp:175-181). */
int ydorb_ba_release(int32_t device);

/* ------------------------------------------------------------------------------------------
 * Pose-only optimisation.  Replaces YDORBSLAM::Optimizer::optimizePose (src/optimizer.cpp:358-501) — the
 * Tracking thread calls it 1-3x per frame right after a match (tracking.cpp:386,466,611).  A batch of frames is solved in one
 * launch, one workgroup per frame running all four episodes.  Frame f owns edges [edge_start[f], edge_start[f+1]): one per
 * keypoint that has a map point; points = MapPoint::getPosInWorld() (world), meas = (kp.x, kp.y, rightX or < 0 for monocular),
 * inv_sigma2 = m_v_invScaleFactorSquares[kp.octave].  poses [n][7] = (t, unit q) of T_c2w, in/out.
 * Outputs: outlier[e] = Frame::m_v_isOutliers of the edge's keypoint, n_inliers[f] = the function's return value
 * (initialCorrespondenceNum - badNum; 0 and an untouched pose when a frame has < 3 edges), optional chi2_log [n][4] (robust chi2
 * after each episode, NaN where an episode did not run) and trials [n] (LM trials executed). */
typedef struct YdPoseBatch {
  int32_t n_frames;
  int32_t device;
  const int32_t* edge_start;     /* [n_frames + 1] */
  double* poses;                 /* [n_frames][7] in/out */
  const double* points;          /* [n_edges][3] */
  const double* meas;            /* [n_edges][3] */
  const double* inv_sigma2;      /* [n_edges] */
  double fx, fy, cx, cy, bf;
} YdPoseBatch;
int ydorb_pose_optimize(const YdPoseBatch* batch, uint8_t* outlier, int32_t* n_inliers, double* chi2_log, int32_t* trials);

/* dense SPD solve with the BA's blocked Cholesky (known-answer tests; A is n x n row-major, host pointers) */
int ydorb_ba_dense_solve(int32_t device, const double* A, int32_t n, const double* b, double* x, int32_t* ok);

#ifdef __cplusplus
}
#endif
#endif /* YDORB_C_API_H */
