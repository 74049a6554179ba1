/* orbhip.h — C ABI of liborbhip.so: the MI355X (gfx950) implementation of ORB-SLAM3's per-frame hot path.
 *
 * The reference has no FFI layer: the boundary is three C++ classes linked into libORB_SLAM3.so
 * (reference CMakeLists.txt:60-118).  Each entry point below names the reference interface it replaces;
 * the headers under include/orbslam3_hip/ hold header-only C++ adapters with the reference's class signatures that
 * forward to this ABI.
 *
 * Conventions: plain pointers and sizes only; caller-allocated outputs with explicit capacity; the library
 * owns device buffers per handle; no ownership crosses the ABI; no exceptions cross the ABI; every function
 * returns ORB_OK (0) or a negative ORB_E_* code, with a message retrievable by orb*_last_error().
 * "dev" pointers are HIP device pointers on the handle's device; `stream` is a hipStream_t (NULL = the HIP
 * default stream; the host-buffer entry points use a private stream of the handle and synchronise it).  Handles are not re-entrant; distinct handles may be used concurrently
 * (reference threading: one ORBextractor per camera, Frame.cc:111-114).
 */
#ifndef ORBHIP_H
#define ORBHIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORB_OK 0
#define ORB_E_EMPTY_IMAGE (-1)   /* mirrors `return -1` of ORBextractor::operator(), ORBextractor.cc:1078-1079 */
#define ORB_E_CAPACITY (-2)      /* caller-provided output capacity too small (n_out holds the needed count) */
#define ORB_E_INVALID (-3)       /* bad argument / unsupported geometry */
#define ORB_E_HIP (-4)           /* HIP runtime error (see last_error) */
#define ORB_E_NOMEM (-5)
#define ORB_E_ABORTED (-6)       /* abort flag was raised (LBA pbStopFlag, Optimizer.cc:2197-2199) */

/* cv::KeyPoint layout (28 bytes): pt.x, pt.y, size, angle, response, octave, class_id */
typedef struct orb_keypoint {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} orb_keypoint;

/* ---------------------------------------------------------------------------------------------------------
 * Stage 1 — ORBextractor  (reference include/ORBextractor.h:49-83, src/ORBextractor.cc)
 * ------------------------------------------------------------------------------------------------------- */
typedef struct orbx_config {
    int32_t nfeatures;     /* ORBextractor.h:49 ctor arg 1 */
    float scale_factor;    /* arg 2 */
    int32_t nlevels;       /* arg 3 */
    int32_t ini_th_fast;   /* arg 4 */
    int32_t min_th_fast;   /* arg 5 */
} orbx_config;

typedef struct orbx_extractor* orbx_handle;

/* Replaces ORBextractor::ORBextractor (ORBextractor.cc:408-468) for images of one fixed size.
 * max_batch: number of frames one orbx_extract_batch_dev call may carry (device workspace is sized for it). */
int orbx_create(const orbx_config* cfg, int width, int height, int max_batch, int device, orbx_handle* out);
void orbx_destroy(orbx_handle h);
const char* orbx_last_error(orbx_handle h); /* h may be NULL: error of the last failed orbx_create */

/* Getters ORBextractor.h:61-81 (GetScaleFactors / GetInverseScaleFactors / GetScaleSigmaSquares /
 * GetInverseScaleSigmaSquares) plus mnFeaturesPerLevel; arrays of nlevels entries, any may be NULL. */
int orbx_get_tables(orbx_handle h, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2,
                    int32_t* features_per_level);
/* Upper bound of keypoints one frame can return (sum over levels of N_l + 3, or the 4*nIni first-round bound). */
int orbx_max_keypoints(orbx_handle h);

/* Replaces ORBextractor::operator() (ORBextractor.cc:1074-1156) for one host image (CV_8UC1, row stride in
 * bytes).  lap0/lap1 = vLappingArea.  Writes n keypoints / n x 32 descriptor bytes in the reference's output
 * order; *mono_index = the reference's return value.  Returns ORB_E_EMPTY_IMAGE for a NULL / zero-sized image. */
int orbx_extract(orbx_handle h, const uint8_t* image, int width, int height, int stride, int lap0, int lap1,
                 orb_keypoint* kps, uint8_t* desc, int cap, int* n_out, int* mono_index);

/* The same call without the two output copies: *kps / *desc point into the handle's pinned output block ([n] records, [n][32] bytes), valid until
 * the next call on the handle.  The adapters copy from there straight into the caller's final containers (std::vector<cv::KeyPoint>, cv::Mat). */
int orbx_extract_view(orbx_handle h, const uint8_t* image, int width, int height, int stride, int lap0, int lap1,
                      const orb_keypoint** kps, const uint8_t** desc, int* n_out, int* mono_index);

/* mvImagePyramid for host consumers (public member ORBextractor.h:83; the only reader in the reference is Frame::ComputeStereoMatches,
 * Frame.cc:962,1052,1071).  keep = 1: every orbx_extract / orbx_extract_view also brings the pyramid of its image to the host in the reference's
 * layout (ORBextractor.cc:1164-1179: each level inside a 19-px BORDER_REFLECT_101 frame) — frame built on the device, ONE copy of the whole slab
 * into pinned memory on a second stream that runs under FAST / octree / describe.  keep = 0 (default): nothing is copied.
 * orbx_host_pyramid_level: pixel (0, 0) of `level` inside that slab (rows `stride` bytes apart, the 19 border pixels lie around it); the
 * storage is the handle's and is overwritten by the next call — cv::Mat headers over it need no per-call allocation. */
int orbx_set_host_pyramid(orbx_handle h, int keep);
int orbx_host_pyramid_level(orbx_handle h, int level, const uint8_t** ptr, int* w, int* hgt, int* stride);

/* Batched, device-resident form of the same call (MI355X addition; frames are independent units).
 * d_images: batch frames, frame b at d_images + b*frame_stride, rows row_stride bytes apart (4-byte aligned).
 * d_kps / d_desc: per-frame slabs of cap_per_frame entries; d_counts[2*b] = n, d_counts[2*b+1] = monoIndex
 * (n > cap_per_frame => that frame's slab holds the first cap_per_frame outputs and n reports the need).
 * Asynchronous on `stream`. */
int orbx_extract_batch_dev(orbx_handle h, const uint8_t* d_images, int batch, size_t frame_stride, int row_stride,
                           int lap0, int lap1, orb_keypoint* d_kps, uint8_t* d_desc, int cap_per_frame,
                           int32_t* d_counts, void* stream);

/* mvImagePyramid (public member, ORBextractor.h:83): device view of level `level` of frame `frame` of the last
 * call (un-bordered plane; the reference's 19-px BORDER_REFLECT_101 frame is produced by orbx_copy_level). */
int orbx_pyramid_level(orbx_handle h, int frame, int level, const uint8_t** d_ptr, int* w, int* hgt, int* stride);
/* Copies level to host; border = 0 (plane) or up to 19 (reference layout incl. reflected frame), out is tightly packed.  Waits for the stream of
 * the call that built the pyramid (never for the whole device); served from the host slab when orbx_set_host_pyramid is on. */
int orbx_copy_level(orbx_handle h, int frame, int level, int border, uint8_t* out);

/* Frame::ComputeStereoMatches (Frame.cc:955-1133) for `batch` rectified stereo pairs whose left / right images were just
 * extracted by `left` / `right` (the 11x11 SAD refinement reads their pyramids — mvImagePyramid of both extractors).
 * kps / desc / counts are the orbx_extract_batch_dev outputs of the two handles (counts = [batch][2]).
 * Outputs per left keypoint: u_right (mvuRight, -1 = no match) and depth (mvDepth, -1); d_work: cap_per_frame int32 per frame. */
int orbx_stereo_matches(orbx_handle left, orbx_handle right, const orb_keypoint* d_kps_l, const uint8_t* d_desc_l,
                        const int32_t* d_counts_l, const orb_keypoint* d_kps_r, const uint8_t* d_desc_r, const int32_t* d_counts_r,
                        int cap_per_frame, int batch, float mb, float mbf, float* d_u_right, float* d_depth, int32_t* d_work,
                        void* stream);

/* Frame::ComputeStereoMatches for the pair whose images `left` and `right` each just took through orbx_extract (Frame.cc:110-114 then :132):
 * key points, descriptors and pyramids are read where those two calls left them on the device — nothing is uploaded; u_right / depth (host,
 * cap entries) receive mvuRight / mvDepth of the *n_left left key points.  Both handles: same device and configuration, last call = orbx_extract. */
int orbx_stereo_matches_last(orbx_handle left, orbx_handle right, float mb, float mbf, float* u_right, float* depth, int cap, int* n_left);

/* Stage-level taps for parity tests (valid after an extract call; host outputs):
 * FAST candidates of (frame, level) = vToDistributeKeys (ORBextractor.cc:776,845-850) as (x,y,score) int triples
 * in arbitrary order (the set is what the octree consumes); returns count via n_out. */
int orbx_debug_candidates(orbx_handle h, int frame, int level, int32_t* xys, int cap, int* n_out);
/* Octree output of (frame, level) in list order (DistributeOctTree result, ORBextractor.cc:737-761):
 * (x,y,score) int triples, coordinates relative to minBorder like the reference at that point. */
int orbx_debug_selected(orbx_handle h, int frame, int level, int32_t* xys, int cap, int* n_out);

/* Measurement facility (bench.py's per-kernel figures).  While enabled, a batch call records five HIP events on its launch stream (they cost the
 * three-stream step 0.5 %: off by default); orbx_last_timing then gives the device time of the last batch call's kernels:
 * ms[0]=pyramid ms[1]=FAST ms[2]=octree ms[3]=orient+blur+rBRIEF ms[4]=total (synchronises the stream; ORB_E_INVALID if no call was timed). */
int orbx_enable_timing(orbx_handle h, int on);
int orbx_last_timing(orbx_handle h, float* ms5);

/* How the last batch call ran FAST (a scheduling decision only: the key points do not depend on it).  two_pass: 1 = detection at iniThFAST followed by a
 * second launch on the cells that came back empty (ORBextractor.cc:812-828), 0 = one pass at min(ini, min).  listed / tiles: the tiles the handle's
 * most recent two-pass call sent to its second pass, of how many — the share that decides whether the next call takes two passes again (copied back
 * asynchronously, both words in one copy: synchronise the stream first for the figure of the call just made; the pair always belongs to ONE call).
 * A HIP graph captured from a batch call keeps the pass form it was captured with (the copy-back is not part of a capture).  Any pointer may be NULL. */
int orbx_last_fast_passes(orbx_handle h, int* two_pass, uint32_t* listed, uint32_t* tiles);

/* Two more scheduling decisions of the last batch call (never the result): frames_ordered = the (frame, tile) / (frame, level) workgroups were handed
 * out heaviest frame first, by the FAST candidate counts the handle's previous call left (a frame that costs ten times the others no longer ends the
 * launch alone); heavy_octree_pass = (frame, level) problems of 8 192 candidates or more were taken by a second k_octree launch with four times the
 * threads (chosen when the call before last had such a problem).  Any pointer may be NULL. */
int orbx_last_schedule(orbx_handle h, int* frames_ordered, int* heavy_octree_pass);

/* ---------------------------------------------------------------------------------------------------------
 * Stage 2 — ORBmatcher  (reference include/ORBmatcher.h:39-94, src/ORBmatcher.cc) + the Frame grid helpers it
 * depends on (src/Frame.cc:444-478, 755-862).  The reference functions walk pointer graphs (Frame&, KeyFrame*,
 * MapPoint*); this ABI works on the flattened records the adapters in include/orbslam3_hip/ORBmatcher.h gather,
 * batched over `batch` independent problems (frames / frame pairs) laid out as fixed-capacity slabs:
 * problem b uses kps + b*cap_k, desc + b*cap_k*32, queries + b*cap_q, ...   All pointers are device pointers.
 * ------------------------------------------------------------------------------------------------------- */
#define ORBM_TH_HIGH 100      /* ORBmatcher.cc:36 */
#define ORBM_TH_LOW 50        /* ORBmatcher.cc:37 */
#define ORBM_HISTO_LENGTH 30  /* ORBmatcher.cc:38 */
#define ORBM_GRID_COLS 64     /* FRAME_GRID_COLS, Frame.h:38 */
#define ORBM_GRID_ROWS 48     /* FRAME_GRID_ROWS, Frame.h:39 */

/* ORBmatcher::DescriptorDistance (ORBmatcher.cc:2700-2716) for every (query, train) pair:
 * out[b][i*nt + j] = Hamming(q[b][i], t[b][j]) as uint16.  nq/nt are the per-problem counts (same for all b). */
int orbm_hamming(const uint8_t* d_q, int nq, const uint8_t* d_t, int nt, int batch, uint16_t* d_out, void* stream);

/* cv::BFMatcher(NORM_HAMMING).knnMatch(k=2) as used by Frame::ComputeStereoFishEyeMatches (Frame.cc:1300):
 * per query the two nearest train descriptors, scanning train indices ascending with strict '<' (equal distances
 * keep the lower train index first).  d_nq / d_nt: per-problem counts (int32, element stride `count_stride`).
 * out_idx / out_dist: [batch][cap_q][2]; missing neighbours are idx -1, dist 256. */
int orbm_knn2(const uint8_t* d_q, const int32_t* d_nq, int cap_q, const uint8_t* d_t, const int32_t* d_nt, int cap_t,
              int count_stride, int batch, int32_t* d_out_idx, int32_t* d_out_dist, void* stream);

/* Frame bounds/grid scalars: mnMinX, mnMinY, mfGridElementWidthInv, mfGridElementHeightInv (Frame.cc:388-399) */
typedef struct orbm_grid_params {
    float min_x, min_y, grid_w_inv, grid_h_inv;
} orbm_grid_params;

/* Frame::AssignFeaturesToGrid + PosInGrid (Frame.cc:444-478, 852-862) as a CSR grid per frame:
 * cell = ix*ORBM_GRID_ROWS + iy; grid_start[b][cell..cell+1] delimit indices into grid_idx[b][], which lists keypoint
 * indices in insertion (ascending) order.  d_nkp: per-frame keypoint counts (element stride count_stride).
 * grid_start: [batch][64*48+1], grid_idx: [batch][cap_k]. */
int orbm_grid_build(const orb_keypoint* d_kps, const int32_t* d_nkp, int count_stride, int cap_k, int batch,
                    const orbm_grid_params* gp, int32_t* d_grid_start, int32_t* d_grid_idx, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Frame constructor steps between the extractor and the matcher (reference src/Frame.cc)
 * ------------------------------------------------------------------------------------------------------- */
/* Pinhole intrinsics (Pinhole::toK) + mDistCoef = (k1, k2, p1, p2, k3) (Frame.h mDistCoef; k3 = 0 for 4-coefficient files) */
typedef struct orbf_camera {
    float fx, fy, cx, cy;
    float dist[5];
} orbf_camera;

/* Frame::UndistortKeyPoints (Frame.cc:874-925): mvKeysUn = mvKeys with pt replaced by cv::undistortPoints(pt, K, mDistCoef, R = I, P = K);
 * dist[0] == 0 copies (:879-883).  In place (d_kps_un == d_kps) is allowed. */
int orbf_undistort_keypoints(const orb_keypoint* d_kps, const int32_t* d_nkp, int count_stride, int cap_k, int batch,
                             const orbf_camera* cam, orb_keypoint* d_kps_un, void* stream);
/* UndistortKeyPoints followed by AssignFeaturesToGrid (the Frame constructor's order, Frame.cc:137-167) as ONE launch, one workgroup per frame:
 * d_kps_un and the CSR grid exactly as orbf_undistort_keypoints + orbm_grid_build write them (single camera; gp from orbf_image_bounds).
 * d_kps_un == d_kps is allowed.  With orbm_enable_timing on, orbm_last_timing's ms[0] is this launch. */
int orbm_undistort_and_grid_build(const orb_keypoint* d_kps, const int32_t* d_nkp, int count_stride, int cap_k, int batch, const orbf_camera* cam,
                                  const orbm_grid_params* gp, orb_keypoint* d_kps_un, int32_t* d_grid_start, int32_t* d_grid_idx, void* stream);
/* Frame::ComputeImageBounds (Frame.cc:926-953) -> bounds = {mnMinX, mnMaxX, mnMinY, mnMaxY} (host), and, if gp != NULL, the grid scalars
 * mfGridElementWidthInv / HeightInv of Frame.cc:394-397 ready for orbm_grid_build.  Synchronous (runs once per calibration). */
int orbf_image_bounds(const orbf_camera* cam, int width, int height, float bounds[4], orbm_grid_params* gp);
/* Frame::ComputeStereoFromRGBD (Frame.cc:1136-1157): depth image(s) float32, frame b at d_depth + b*frame_stride, rows row_stride floats
 * apart; mvDepth = d, mvuRight = kpU.x - mbf/d where d > 0, else -1.  Entries [n, cap_k) of the outputs are set to -1. */
int orbf_stereo_from_rgbd(const orb_keypoint* d_kps, const orb_keypoint* d_kps_un, const int32_t* d_nkp, int count_stride, int cap_k,
                          int batch, const float* d_depth, size_t frame_stride, int row_stride, int width, int height, float mbf,
                          float* d_u_right, float* d_depth_out, void* stream);

/* Frame::ComputeStereoFishEyeMatches (Frame.cc:1281-1325) with KannalaBrandt8::TriangulateMatches (KannalaBrandt8.cpp:334-400) for two fisheye
 * cameras.  Left / right keypoints and descriptors as the two extractors wrote them; d_n_* = Nleft / Nright, d_mono_* = monoLeft / monoRight (the
 * extractors' return values: keypoints [mono, N) lie in the lapping area), element stride count_stride (2 for orbx_extract_batch_dev counts:
 * pass d_counts and d_counts + 1).  Outputs: mvLeftToRightMatch [batch][cap_l], mvRightToLeftMatch [batch][cap_r] (indices into the own camera's
 * arrays, -1 = none), mvDepth [batch][cap_l] (-1), mvStereo3Dpoints [batch][cap_l][3] (left-camera frame; zeros where none), nMatches [batch]. */
typedef struct orbf_fisheye_rig {
    float k_left[8], k_right[8];   /* KannalaBrandt8 mvParameters fx fy cx cy k1..k4 of mpCamera / mpCamera2 */
    float R_lr[9], t_lr[3];        /* mRlr (row-major), mtlr = the rotation / translation of mTlr (Frame.cc:1242-1243) */
    float level_sigma2[16];        /* mvLevelSigma2 */
} orbf_fisheye_rig;
int orbf_stereo_fisheye_matches(const orb_keypoint* d_kps_l, const uint8_t* d_desc_l, const int32_t* d_n_l, const int32_t* d_mono_l,
                                const orb_keypoint* d_kps_r, const uint8_t* d_desc_r, const int32_t* d_n_r, const int32_t* d_mono_r, int cap_l,
                                int cap_r, int count_stride, int batch, const orbf_fisheye_rig* rig, int32_t* d_left_to_right,
                                int32_t* d_right_to_left, float* d_depth, float* d_p3d, int32_t* d_nmatches, void* stream);

/* One projected map point = one query of a windowed search (the per-MapPoint values the reference computes before
 * calling Frame::GetFeaturesInArea: ORBmatcher.cc:88-103 for the local-map search, :2277-2309 for the motion model). */
typedef struct orbm_query {
    float u, v;          /* projection (mTrackProjX/Y, or uv) */
    float radius;        /* r * mvScaleFactors[level]  (window half-size, also the stereo gate) */
    float u_right;       /* mTrackProjXR / uv.x - mbf*invzc; only read when ORBM_Q_STEREO is set */
    float angle;         /* keypoint angle of the source observation (rotation histogram), degrees */
    int16_t min_level, max_level;   /* GetFeaturesInArea level window (its bCheckLevels rule is applied as is) */
    uint32_t flags;
} orbm_query;
#define ORBM_Q_VALID 1u      /* mbTrackInView && !isBad ... : query takes part */
#define ORBM_Q_STEREO 2u     /* apply the |u_right - mvuRight[idx]| <= radius gate where mvuRight[idx] > 0 */
#define ORBM_Q_HAS_OBS 4u    /* pMP->Observations() > 0: once matched, the keypoint is skipped by later queries */
#define ORBM_Q_RIGHT 8u      /* fisheye rig: search the right camera's grid (GetFeaturesInArea(..., bRight = true)) */
#define ORBM_Q_TWIN 16u      /* fisheye rig: right-camera query of the same map point as the previous query */

typedef struct orbm_search_params {
    int32_t mode;              /* ORBM_MODE_LOCAL_MAP, ORBM_MODE_BEST_ONLY or ORBM_MODE_INIT */
    int32_t th_dist;           /* TH_HIGH (100), or ORBdist of the relocalisation variant */
    float nn_ratio;            /* mfNNratio (LOCAL_MAP only) */
    int32_t check_orientation; /* mbCheckOrientation: rotation-histogram cull (BEST_ONLY only, as in the reference) */
    orbm_grid_params grid;
} orbm_search_params;
#define ORBM_MODE_LOCAL_MAP 0  /* SearchByProjection(Frame&, vector<MapPoint*>&, th, ...)  ORBmatcher.cc:59-255 (left camera) */
#define ORBM_MODE_BEST_ONLY 1  /* SearchByProjection(Frame&, const Frame&, th, bMono)      ORBmatcher.cc:2244-2509;
                                * SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist) :2520-2652 with th_dist = ORBdist,
                                *   occupied0 = (mvpMapPoints[i] != NULL), every query ORBM_Q_HAS_OBS;
                                * SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th, ratioHamming) :593-824 with
                                *   check_orientation = 0, th_dist = floor(TH_LOW*ratioHamming), levels [L-1, L] */

#define ORBM_MODE_INIT 2       /* SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize)  ORBmatcher.cc:838-979:
                                *   one query per F1 keypoint (VALID iff octave == 0; u,v = vbPrevMatched[i1]; radius = windowSize;
                                *   min_level = max_level = 0; angle = F1 keypoint angle), th_dist = TH_LOW, nn_ratio, check_orientation.
                                *   A candidate is skipped while vMatchedDistance[i2] <= dist; an accepted match displaces the previous
                                *   holder of i2.  q_match = vnMatches12, kp_match = vnMatches21 (before the orientation cull).
                                *   Histogram factor HISTO_LENGTH/360 (this fork, :852).  cap_q <= 65534. */

/* Windowed projection search, results identical to the reference's serial loop (queries are resolved in index
 * order; a keypoint claimed by an earlier query with ORBM_Q_HAS_OBS is skipped by later ones).
 * Frame side: kps (x,y,octave,angle of mvKeysUn), desc, u_right (mvuRight, may be NULL = mono), occupied0 (may be NULL;
 * non-zero = mvpMapPoints[idx] already holds an observed point), CSR grid from orbm_grid_build.
 * Outputs: q_match[b][q] = matched keypoint index or -1; kp_match[b][idx] = index of the query whose map point
 * mvpMapPoints[idx] holds after the call, -1 = untouched by the call (mvpMapPoints[idx] keeps its value), -2 = claimed during the
 * call and then set to NULL by the orientation cull (ORBmatcher.cc:2499, :2637); nmatches[b] = the reference's return value.
 * d_work: scratch of orbm_search_workspace_bytes(batch, cap_q) bytes.
 * Form: the occupancy modes (LOCAL_MAP, BEST_ONLY) on single-camera frames with cap_q <= 2048 run as ONE workgroup per frame — window walk from an LDS copy of the
 * frame, the serial accept loop as parallel fixed-point rounds: the same result by construction (DESIGN.md stage-2 notes); every other call as per-query candidate
 * lists followed by a one-wave walk in query order. */
size_t orbm_search_workspace_bytes(int batch, int cap_q);
int orbm_search_by_projection(const orb_keypoint* d_kps, const uint8_t* d_desc, const float* d_u_right, const uint8_t* d_occupied0,
                              const int32_t* d_nkp, int count_stride, int cap_k,
                              const int32_t* d_grid_start, const int32_t* d_grid_idx,
                              const orbm_query* d_queries, const uint8_t* d_qdesc, const int32_t* d_nq, int cap_q, int batch,
                              const orbm_search_params* params, int32_t* d_q_match, int32_t* d_kp_match, int32_t* d_nmatches,
                              void* d_work, void* stream);

/* Measurement facility (bench.py's roofline legs), the stage-2 counterpart of orbx_last_timing: while enabled, HIP events are recorded on the launch
 * stream around the kernels of orbm_grid_build and orbm_search_by_projection; orbm_last_timing synchronises on them and returns the device time of the
 * last call's kernels: ms[0] = grid build (or the fused undistort + grid launch), ms[1] = the projection search of the frames in one workgroup each (k_sbp_frame; calls it does not
 * cover — INIT mode, rigs, cap_q > 2048 —: candidate enumeration + Hamming), ms[2] = the gated launches behind it for flagged frames (those calls: the serial-order resolution).  The switch and the events are per calling thread: a thread times its own calls only, concurrent matcher
 * calls of other threads neither record into nor disturb them — so orbm_enable_timing, the timed calls and orbm_last_timing must come from the SAME thread
 * (another thread reads zeros), on one device (the events are re-made, and the last figures dropped, when the thread's current device changes). */
int orbm_enable_timing(int on);
int orbm_last_timing(float* ms3);

/* ORBmatcher::SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th) (ORBmatcher.cc:2008-2220) = orbm_fuse in both directions (each map point
 * keeps its own best candidate in [L-1, L] with bestDist <= TH_HIGH and no chi2 gate: vnMatch1 / vnMatch2, :2044-2119 and :2122-2201) followed by
 * this agreement pass (:2203-2219): out12[b][i1] = idx2 iff match12[b][i1] == idx2 and match21[b][idx2] == i1, else -1; nfound[b] = the return value. */
int orbm_mutual_matches(const int32_t* d_match12, const int32_t* d_match21, const int32_t* d_n1, const int32_t* d_n2, int cap1, int cap2,
                        int batch, int32_t* d_out12, int32_t* d_nfound, void* stream);

/* Fisheye-rig (F.Nleft != -1) variants.  Keypoints / descriptors are the concatenation [mvKeys | mvKeysRight] like the reference's
 * N-sized arrays; d_nleft[b] = Nleft.  The grid has 2 x 64 x 48 cells per frame (second half = mGridRight, entries are global indices):
 * grid_start [batch][2*64*48+1].  d_kp_link[b][i] = global index of keypoint i's stereo partner (mvLeftToRightMatch[i] + Nleft, or
 * mvRightToLeftMatch[i - Nleft]) or -1; may be NULL.  A map point seen by both cameras contributes two consecutive queries: the left
 * one, then the right one flagged ORBM_Q_RIGHT | ORBM_Q_TWIN.  Reproduced quirks:
 *   ORBM_MODE_LOCAL_MAP (ORBmatcher.cc:59-258): the `continue` of the left ratio test (:166-167) also skips the right camera; an
 *     accepted match is copied to the stereo partner and counted twice (:172-176, :239-243); u_right gating does not apply to rigs;
 *   ORBM_MODE_BEST_ONLY (:2244-2509): an empty left window (`if(vIndices2.empty()) continue;` :2332) or an invalid left query skips
 *     the right camera of that map point. */
int orbm_grid_build_rig(const orb_keypoint* d_kps, const int32_t* d_nkp, const int32_t* d_nleft, int count_stride, int cap_k, int batch,
                        const orbm_grid_params* gp, int32_t* d_grid_start, int32_t* d_grid_idx, void* stream);
int orbm_search_by_projection_rig(const orb_keypoint* d_kps, const uint8_t* d_desc, const uint8_t* d_occupied0, const int32_t* d_kp_link,
                                  const int32_t* d_nkp, int count_stride, int cap_k,
                                  const int32_t* d_grid_start, const int32_t* d_grid_idx,
                                  const orbm_query* d_queries, const uint8_t* d_qdesc, const int32_t* d_nq, int cap_q, int batch,
                                  const orbm_search_params* params, int32_t* d_q_match, int32_t* d_kp_match, int32_t* d_nmatches,
                                  void* d_work, void* stream);

/* ORBmatcher::Fuse — the search half of both overloads: per projected map point the best keypoint of the key frame.
 *   chi2_gate = 1: Fuse(KeyFrame*, const vector<MapPoint*>&, th, bRight)           ORBmatcher.cc:1630-1882 (search :1770-1830)
 *   chi2_gate = 0: Fuse(KeyFrame*, cv::Mat Scw, vpPoints, th, vpReplacePoint)      ORBmatcher.cc:1884-2006 (search :1960-1985)
 * A query = what the reference computes per map point before KeyFrame::GetFeaturesInArea (KeyFrame.cc:810-854): u,v = uv,
 * u_right = uv.x - bf*invz, radius = th*mvScaleFactors[nPredictedLevel], min_level = nPredictedLevel-1, max_level = nPredictedLevel,
 * flags = ORBM_Q_VALID iff the point passed the frustum / distance / normal gates (:1700-1765).  Queries are independent (several map
 * points may select the same keypoint; the Replace / AddObservation mutations stay in the host adapter, applied in index order).
 * Outputs: q_match[b][q] = bestIdx if bestDist <= th_dist else -1; q_dist[b][q] = bestDist (256: empty window); nfused[b]. */
typedef struct orbm_fuse_params {
    int32_t th_dist;              /* TH_LOW */
    int32_t chi2_gate;
    orbm_grid_params grid;
    float inv_level_sigma2[16];   /* pKF->mvInvLevelSigma2 (chi2_gate only) */
} orbm_fuse_params;
int orbm_fuse(const orb_keypoint* d_kps, const uint8_t* d_desc, const float* d_u_right, const int32_t* d_nkp, int count_stride, int cap_k,
              const int32_t* d_grid_start, const int32_t* d_grid_idx, const orbm_query* d_queries, const uint8_t* d_qdesc,
              const int32_t* d_nq, int cap_q, int batch, const orbm_fuse_params* params, int32_t* d_q_match, int32_t* d_q_dist,
              int32_t* d_nfused, void* stream);

/* ORBmatcher::SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo, bCoarse) (ORBmatcher.cc:1138-1428; call sites
 * LocalMapping.cc:628, Tracking.cc:3812) for pinhole key frames without a second camera.  Both FeatureVectors as CSR like
 * orbm_bow_side; has_mp[i] != 0 where pKF->GetMapPoint(i) != NULL; u_right = mvuRight (NULL = monocular).
 * Per pair: the fundamental matrix Pinhole::epipolarConstrain builds (Pinhole.cpp:157-160, row-major F12 = K1^-T [t12]x R12 K2^-1;
 * the adapter evaluates that cv::Mat expression once per pair), the epipole ep (ORBmatcher.cc:1152) and pKF2's mvLevelSigma2 /
 * mvScaleFactors.  Output match12[b][i1] = idx2 or -1 (vMatches12 after the orientation cull), nmatches[b]. */
typedef struct orbm_tri_side {
    const orb_keypoint* kps;      /* [batch][cap_f]  mvKeysUn */
    const uint8_t* desc;          /* [batch][cap_f][32] */
    const float* u_right;         /* [batch][cap_f] or NULL */
    const uint8_t* has_mp;        /* [batch][cap_f] */
    const int32_t* node_id;       /* [batch][cap_nodes] ascending */
    const int32_t* node_start;    /* [batch][cap_nodes+1] */
    const int32_t* feat_idx;      /* [batch][cap_f] */
    const int32_t* n_nodes;       /* [batch] */
    int32_t cap_f, cap_nodes;
} orbm_tri_side;
typedef struct orbm_tri_pair {
    float F12[9];
    float ep[2];
    float level_sigma2_2[16];     /* pKF2->mvLevelSigma2 */
    float scale_factors_2[16];    /* pKF2->mvScaleFactors */
    float reserved;
} orbm_tri_pair;
int orbm_search_for_triangulation(const orbm_tri_side* kf1, const orbm_tri_side* kf2, const orbm_tri_pair* d_pairs, int batch,
                                  int only_stereo, int coarse, int check_orientation, int32_t* d_match12, int32_t* d_nmatches,
                                  void* stream);

/* The same search for key frames whose cameras are KannalaBrandt8 (BASELINE configs[3]): a monocular fisheye camera
 * (n_cams = 1) or a fisheye rig with mpCamera2 (n_cams = 2; kps / desc = the concatenation [mvKeys | mvKeysRight] like a rig Frame — NOT
 * mvKeysUn, ORBmatcher.cc:1249-1251 — with d_nleft*[b] = NLeft, features >= NLeft belong to the right camera).  The gate is
 * KannalaBrandt8::epipolarConstrain = TriangulateMatches(pCamera2, kp1, kp2, R12, t12, mvLevelSigma2_1[kp1.octave], mvLevelSigma2_2[kp2.octave]) >
 * 0.0001f (KannalaBrandt8.cpp:235-238, 334-400) with the (camera, R12, t12) combination chosen per candidate from (bRight1, bRight2)
 * (ORBmatcher.cc:1280-1315).  bStereo1 / bStereo2 are false for such key frames (u_right of the sides is ignored), so only_stereo yields no
 * match; the epipole test (:1269-1277) applies only when n_cams = 1.  The float32 cv::Mat / libm / cv::SVD steps inside TriangulateMatches
 * follow rule R4, like orbf_stereo_fisheye_matches: parity vs a real OpenCV build is unpinned for them. */
typedef struct orbm_tri_kb8_pair {
    int32_t n_cams, reserved;
    float k1[2][8], k2[2][8];     /* mvParameters of pKF1->mpCamera / mpCamera2 and of pKF2->mpCamera / mpCamera2 */
    float R12[4][9], t12[4][3];   /* row-major, index bRight1*2 + bRight2: Rll, Rlr, Rrl, Rrr / tll, tlr, trl, trr (ORBmatcher.cc:1181-1193) as the adapter's
                                   * cv::Mat expressions evaluate them; n_cams = 1: entry 0 = R12, t12 (:1176-1177) */
    float ep[2];                  /* pKF2->mpCamera->project(R2w*Cw + t2w) (:1149-1152); read only when n_cams = 1 */
    float level_sigma2_1[16], level_sigma2_2[16], scale_factors_2[16];   /* pKF1->mvLevelSigma2, pKF2->mvLevelSigma2, pKF2->mvScaleFactors */
} orbm_tri_kb8_pair;
int orbm_search_for_triangulation_kb8(const orbm_tri_side* kf1, const orbm_tri_side* kf2, const int32_t* d_nleft1, const int32_t* d_nleft2,
                                      const orbm_tri_kb8_pair* d_pairs, int batch, int only_stereo, int coarse, int check_orientation,
                                      int32_t* d_match12, int32_t* d_nmatches, void* stream);

/* ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vpMapPointMatches) (ORBmatcher.cc:323-587, Nleft == -1 branch).
 * FeatureVector of each side as CSR: node ids ascending (std::map order), node_start[k..k+1] delimit feat_idx[] (the
 * feature indices of that node in insertion order).  kf_valid[i] != 0 where the KF feature holds a good map point.
 * Output f_match[b][j] = KF feature index matched to frame feature j or -1 (vpMapPointMatches), nmatches[b]. */
typedef struct orbm_bow_side {
    const uint8_t* desc;          /* [batch][cap_f][32] */
    const float* angle;           /* [batch][cap_f]  keypoint angles */
    const int32_t* node_id;       /* [batch][cap_nodes] ascending */
    const int32_t* node_start;    /* [batch][cap_nodes+1] */
    const int32_t* feat_idx;      /* [batch][cap_f] */
    const int32_t* n_nodes;       /* [batch] */
    int32_t cap_f, cap_nodes;
    const int32_t* n_left;        /* frame side only: [batch] F.Nleft of a fisheye rig (features >= Nleft belong to the right camera), or
                                   * NULL / -1 for a single camera.  Rig frames take the `F.Nleft != -1` branch of ORBmatcher.cc:411-436:
                                   * separate best/second for the left and the right features of a node; the right match is accepted
                                   * whenever the left best passed TH_LOW and bestDist1R <= TH_LOW (the `|| true` of :509) */
} orbm_bow_side;
int orbm_search_by_bow(const orbm_bow_side* kf, const uint8_t* d_kf_valid, const orbm_bow_side* f, int batch,
                       float nn_ratio, int check_orientation, int32_t* d_f_match, int32_t* d_nmatches, void* stream);

/* ORBmatcher::SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, vector<MapPoint*>& vpMatches12) (ORBmatcher.cc:984-1124; call site
 * LoopClosing.cc:697).  Both FeatureVectors as CSR (n_left of the sides is ignored).  valid1[i] / valid2[i] != 0 where the key frame's
 * feature i holds a map point that is not bad (:1025-1028, :1049-1053) and, for a fisheye-rig key frame (NLeft != -1), i < mvKeysUn.size()
 * (:1020-1022, :1043-1045) — the adapter folds both conditions into the flag.  Output match12[b][i1] = index of the pKF2 feature whose
 * map point vpMatches12[i1] holds, or -1 ([batch][kf1->cap_f]); nmatches[b] = the return value.  Unlike the (KeyFrame, Frame) overload the
 * acceptance is `bestDist1 < TH_LOW` (strict, :1072) and a pKF2 feature is taken at most once (vbMatched2, :1077). */
int orbm_search_by_bow_kf(const orbm_bow_side* kf1, const uint8_t* d_valid1, const orbm_bow_side* kf2, const uint8_t* d_valid2, int batch,
                          float nn_ratio, int check_orientation, int32_t* d_match12, int32_t* d_nmatches, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * SURVEY.md N2 — the DBoW2 step between the extractor and SearchByBoW: Frame::ComputeBoW (reference src/Frame.cc:865-872) =
 * TemplatedVocabulary::transform(features, BowVector&, FeatureVector&, levelsup = 4)
 * (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1137-1206; per-descriptor tree descent :1231-1272 with FORB::distance, FORB.cpp:81-101),
 * on a vocabulary in the binary format System.cc:83 loads (loadFromBinaryFile, TemplatedVocabulary.h:1442-1480:
 * header u32 nb_nodes, u32 size_node, i32 k, i32 L, i32 scoring, i32 weighting; then nb_nodes records
 * {i32 parent, 32 B descriptor, f32 weight, u8 is_leaf}; node ids 1..nb_nodes in file order, word ids in leaf order).
 * ------------------------------------------------------------------------------------------------------- */
typedef struct bow_vocab* bow_vocab_handle;
/* Parses the file image (host memory) and uploads the tree.  ORB_E_INVALID for a short or inconsistent file. */
int bow_vocab_load_binary(const void* file_bytes, size_t n_bytes, int device, bow_vocab_handle* out);
/* out6 = {k, L, scoring (DBoW2::ScoringType), weighting (DBoW2::WeightingType), nodes (without the root), words} */
int bow_vocab_info(bow_vocab_handle v, int32_t* out6);
void bow_vocab_destroy(bow_vocab_handle v);

/* Outputs of one transform() per frame, fixed-capacity slabs (device pointers; frame b uses x + b*cap_f, ...):
 *   per feature i (what transform(feature, id, w, &nid, levelsup) returns): word_id, node_id (NodeId at level L - levelsup), weight;
 *   FeatureVector (std::map<NodeId, vector<uint>>) as the CSR orbm_search_by_bow / orbm_search_for_triangulation consume:
 *     fv_node_id[b][cap_f] ascending, fv_node_start[b][cap_f+1], fv_feat_idx[b][cap_f] (ascending inside a node), fv_n_nodes[b];
 *   BowVector (std::map<WordId, double>) as bv_word[b][cap_f] ascending, bv_value[b][cap_f], bv_n[b] — summed / normalised in
 *     the reference's order (BowVector.cpp:34-84), so the doubles are bit-identical.
 * Features whose word weight is 0 ("stopped") appear in neither vector.  cap_f <= 4096. */
typedef struct bow_result {
    int32_t* word_id; int32_t* node_id; double* weight;
    int32_t* fv_node_id; int32_t* fv_node_start; int32_t* fv_feat_idx; int32_t* fv_n_nodes;
    int32_t* bv_word; double* bv_value; int32_t* bv_n;
} bow_result;
int bow_transform(bow_vocab_handle v, const uint8_t* d_desc, const int32_t* d_n, int count_stride, int cap_f, int batch, int levelsup,
                  const bow_result* out, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Stage 3 — Optimizer::LocalBundleAdjustment's linearisation  (reference src/Optimizer.cc:1957-2344 graph,
 * src/OptimizableTypes.{h,cpp} edges, Thirdparty/g2o core/block_solver.hpp:502-560 buildSystem,
 * core/base_binary_edge.hpp:55-120 constructQuadraticForm, core/robust_kernel_impl.cpp:78-91 Huber).
 * The adapter (include/orbslam3_hip/Optimizer.h) flattens the g2o graph into the SoA below once per optimize()
 * call (== BlockSolver::buildStructure) and then asks for one linearisation per LM iteration / trial.
 * `batch` independent windows use fixed-capacity slabs like stage 2; all pointers are device pointers, all
 * matrices are column-major doubles as in g2o's Eigen::Map blocks.
 * ------------------------------------------------------------------------------------------------------- */
#define LBA_EDGE_MONO 0    /* ORB_SLAM3::EdgeSE3ProjectXYZ        OptimizableTypes.h:102-130, .cpp:142-172 */
#define LBA_EDGE_STEREO 1  /* g2o::EdgeStereoSE3ProjectXYZ        types_six_dof_expmap.h:146-175, .cpp:190-275 */
#define LBA_EDGE_BODY 2    /* ORB_SLAM3::EdgeSE3ProjectXYZToBody  OptimizableTypes.h:132-159, .cpp:204-225 */
#define LBA_CAM_PINHOLE 0  /* CameraModels/Pinhole.cpp:43-49, 89-100 */
#define LBA_CAM_KB8 1      /* CameraModels/KannalaBrandt8.cpp:52-66, 166-196 */

typedef struct lba_camera {
    int32_t model, reserved;
    double p[8];        /* mvParameters (float values widened): fx fy cx cy [k1 k2 k3 k4] */
    double bf;          /* stereo baseline * fx (EdgeStereoSE3ProjectXYZ::bf) */
    double trl_q[4];    /* mTrl rotation as quaternion x y z w (body edges) */
    double trl_t[3];    /* mTrl translation */
} lba_camera;

typedef struct lba_edge {      /* one observation; edges are stored landmark-major (Optimizer.cc:2060-2190 insertion order) */
    int32_t pose, point;       /* indices into the window's pose / point arrays */
    int16_t kind, cam;         /* LBA_EDGE_*, index into cameras */
    float obs[3];              /* kpUn.pt.x, kpUn.pt.y, mvuRight (stereo) — float in the map, widened on use */
    float inv_sigma2;          /* mvInvLevelSigma2[octave]; information = I * inv_sigma2 */
} lba_edge;

typedef struct lba_problem {
    /* vertices */
    const double* poses;        /* [batch][cap_p][7]  SE3Quat estimate: t(x y z), q(x y z w) */
    const int32_t* pose_hidx;   /* [batch][cap_p]     Hessian block index of the pose (ascending KF id), -1 = fixed vertex */
    const double* points;       /* [batch][cap_l][3]  VertexSBAPointXYZ estimates */
    /* edges + the two CSR views built once per optimize() (BlockSolver::buildStructure, block_solver.hpp:143-295) */
    const lba_edge* edges;      /* [batch][cap_e] */
    const int32_t* lm_start;    /* [batch][cap_l+1]   edges of landmark l are [lm_start[l], lm_start[l+1]) */
    const int32_t* pose_start;  /* [batch][cap_p+1]   CSR over pose_edges */
    const int32_t* pose_edges;  /* [batch][cap_e]     edge indices of each pose, ascending */
    const lba_camera* cameras;  /* [n_cameras] shared by the batch */
    const int32_t* n_poses;     /* [batch] */
    const int32_t* n_points;    /* [batch] */
    const int32_t* n_edges;     /* [batch] */
    int32_t cap_p, cap_l, cap_e, n_cameras;
    double huber_mono, huber_stereo;   /* thHuberMono = sqrt(5.991), thHuberStereo = sqrt(7.815); <= 0 disables the kernel */
} lba_problem;

typedef struct lba_system {     /* outputs; any pointer may be NULL to skip that product */
    double* Hpp;     /* [batch][cap_p][36]  6x6 block of free pose hidx (rotation first, then translation) */
    double* bp;      /* [batch][cap_p][6] */
    double* Hll;     /* [batch][cap_l][9] */
    double* bl;      /* [batch][cap_l][3] */
    double* Hpl;     /* [batch][cap_e][18]  logical 6x3 block H(pose, landmark) of each edge with a free pose (else zeros) */
    double* err;     /* [batch][cap_e][3]   _error = obs - projection */
    double* chi2;    /* [batch][cap_e]      e^T Omega e */
    double* rho;     /* [batch][cap_e][2]   Huber rho[0], rho[1] */
    double* depth;   /* [batch][cap_e]      z of the point in the (projecting) camera frame: isDepthPositive() */
    double* robust_chi2_sum;  /* [batch]    sum of rho[0] = SparseOptimizer::activeRobustChi2 (sparse_optimizer.cpp:100-114) */
} lba_system;

/* BlockSolver::buildSystem (block_solver.hpp:502-560): linearizeOplus + constructQuadraticForm over all active edges. */
int lba_build_system(const lba_problem* prob, int batch, const lba_system* out, void* stream);
/* The same with what the caller knows about the graph it flattened.  LBA_HINT_MONO_PINHOLE: EVERY edge of the batch is an EdgeSE3ProjectXYZ
 * (LBA_EDGE_MONO) on a pinhole camera — a monocular pinhole map (Optimizer.cc:2095-2127 took the `mono` branch for every observation).  The
 * kernels specialised for it give identical results with a third fewer registers; a hint that does not hold gives wrong blocks (lba_optimize
 * checks the edges itself, once per call, and needs no hint).  Unknown hint bits: ORB_E_INVALID. */
#define LBA_HINT_MONO_PINHOLE 1u
#define LBA_HINT_PINHOLE 2u        /* every edge is LBA_EDGE_MONO or LBA_EDGE_STEREO on a pinhole camera (stereo / RGB-D pinhole maps); also for pose_optimize_hint */
int lba_build_system_hint(const lba_problem* prob, int batch, const lba_system* out, unsigned hints, void* stream);
/* SparseOptimizer::computeActiveErrors (sparse_optimizer.cpp:61-75): err / chi2 / rho / depth / robust_chi2_sum only. */
int lba_compute_errors(const lba_problem* prob, int batch, const lba_system* out, void* stream);

/* SURVEY.md N4 — the rest of one SparseOptimizer::optimize(iterations) call on the LBA graph, GPU resident:
 * OptimizationAlgorithmLevenberg::solve (g2o/core/optimization_algorithm_levenberg.cpp:61-168: _tau = 1e-50, <= 100 lambda
 * trials, rho test, the "nBad" stop), BlockSolver::setLambda / Schur complement / back-substitution
 * (block_solver.hpp:564-589, 381-481) and the vertex updates (exp(dx)*T, X += dx).  The reduced camera system is solved
 * with a dense Cholesky per window (g2o: Eigen SimplicialLDLT — same solution of the SPD system).
 * prob->poses and prob->points are UPDATED IN PLACE (they must be writable device memory).  h_stats: [batch][4] host doubles =
 * {iterations run, final activeRobustChi2, final lambda, total lambda trials}.  abort_flag (may be NULL) is polled between
 * trials like pbStopFlag (Optimizer.cc:2197-2199, sparse_optimizer.cpp:376) -> ORB_E_ABORTED.  Synchronous on `stream`. */
size_t lba_lm_workspace_bytes(const lba_problem* prob, int batch);
int lba_optimize(const lba_problem* prob, int batch, int iterations, void* d_workspace, double* h_stats,
                 const volatile int* abort_flag, void* stream);
/* The same call polling the reference's own stop flag: Optimizer::LocalBundleAdjustment gets `bool* pbStopFlag` (one byte, set by LocalMapping from
 * another thread) and passes it to g2o's setForceStopFlag (Optimizer.cc:1975-1976); pb_stop_flag may be NULL. */
int lba_optimize_stopflag(const lba_problem* prob, int batch, int iterations, void* d_workspace, double* h_stats,
                          const volatile unsigned char* pb_stop_flag, void* stream);

/* lba_optimize for windows SHARDED BY LANDMARK over several processes, one per GPU (BASELINE configs[4]; the reference solves one
 * window in one process: BlockSolver::buildSystem / solve, block_solver.hpp:381-432,502-560).  Every rank passes the whole pose set of each window and the
 * points / edges of ITS landmarks only (re-indexed from 0).  Per linearisation the pose-side blocks H_pp | b_p are summed over the ranks, per lambda
 * trial the reduced camera system [np6 x np6 | np6] and three scalars per window (chi2, computeScale, failures), once the lambda start value (max):
 * all through `reduce(user, d_buf, n, op, stream)`, which must leave d_buf[0..n) = the sum (op 0) or maximum (op 1) over all ranks of their d_buf,
 * bit-identical on every rank, ordered after the work already queued on `stream` and complete (or stream-ordered) when it returns; non-zero = failure.
 * Every rank then takes the same decisions, factorises the same system and updates ALL poses; its own landmarks are back-substituted locally.
 * owner: non-zero on exactly one rank (the terms that exist once — H_pp + lambda I, b_p, the pose part of computeScale — enter the sums there).
 * h_stats as lba_optimize, identical on every rank (chi2 of the WHOLE window).  There is no stop flag: the ranks must run the same trials. */
typedef int (*lba_allreduce_fn)(void* user, double* d_buf, size_t n, int op, void* stream);
int lba_optimize_sharded(const lba_problem* prob, int batch, int iterations, void* d_workspace, double* h_stats, int owner,
                         lba_allreduce_fn reduce, void* user, void* stream);
/* lba_optimize returns ORB_E_CAPACITY when the reduced camera system (6 unknowns per FREE key frame; fixed key frames do not count) no
 * longer fits the solver's LDS budget (> ~3 300 free key frames): nothing was changed, the window stays as it was.  A host that must optimise such a map
 * keeps its own CPU solver for it (integration/Optimizer_hip.cc REPLACES the g2o body and therefore leaves the map untouched instead — GlueGuard.h). */

/* SURVEY.md N3 — Optimizer::PoseOptimization(Frame*) (reference src/Optimizer.cc:907-1273): the motion-only BA that follows
 * every matcher call in Tracking (Tracking.cc:2210,2395,2468).  One pose vertex, unary reprojection edges
 * (EdgeSE3ProjectXYZOnlyPose OptimizableTypes.h:37-63/.cpp:50-65, EdgeSE3ProjectXYZOnlyPoseToBody :65-91/.cpp:93-109,
 * g2o::EdgeStereoSE3ProjectXYZOnlyPose types_six_dof_expmap.cpp:339-404), 4 rounds x optimize(10) of g2o's Levenberg-Marquardt
 * with the dense 6x6 solver, chi2 inlier/outlier re-classification between rounds (5.991 / 7.815), Huber kernel dropped for the
 * last round, pose reset to the initial estimate at the start of every round.  One workgroup per frame runs the whole thing
 * in a single launch.  Outputs: the optimised pose, mvbOutlier per edge, and the return value nInitialCorrespondences - nBad. */
typedef struct pose_edge {
    float xw[3];        /* pMP->GetWorldPos() (float in the map) */
    float obs[3];       /* kpUn.pt.x, kpUn.pt.y, mvuRight */
    float inv_sigma2;   /* mvInvLevelSigma2[octave] */
    int16_t kind, cam;  /* LBA_EDGE_MONO / LBA_EDGE_STEREO / LBA_EDGE_BODY, camera index */
} pose_edge;
int pose_optimize(const double* d_poses_in, const pose_edge* d_edges, const int32_t* d_n_edges, int cap_e, int batch,
                  const lba_camera* d_cameras, int n_cameras, double* d_poses_out, uint8_t* d_outlier, int32_t* d_n_good,
                  void* stream);
/* The same with what the caller knows about the edges it flattened.  LBA_HINT_PINHOLE: EVERY edge is an EdgeSE3ProjectXYZOnlyPose (LBA_EDGE_MONO)
 * or an EdgeStereoSE3ProjectXYZOnlyPose (LBA_EDGE_STEREO) on a pinhole camera — the monocular, stereo and RGB-D pinhole configurations.
 * Identical results from a kernel without the fisheye model and the right-camera edge; a hint that does not hold gives wrong poses. */
int pose_optimize_hint(const double* d_poses_in, const pose_edge* d_edges, const int32_t* d_n_edges, int cap_e, int batch,
                       const lba_camera* d_cameras, int n_cameras, double* d_poses_out, uint8_t* d_outlier, int32_t* d_n_good,
                       unsigned hints, void* stream);

/* SURVEY.md N4 (tail) — Optimizer::LocalInertialBA (reference src/Optimizer.cc:4753-5365): the visual-inertial local BA of the
 * inertial modes.  Graph (include/G2oTypes.h, src/G2oTypes.cc): VertexPose (ImuCamPose, 6 dof, body-frame right perturbation
 * G2oTypes.cc:196-221), VertexVelocity / VertexGyroBias / VertexAccBias (3 dof each, additive), VertexSBAPointXYZ (marginalised);
 * EdgeInertial (9-dim, six vertices, G2oTypes.cc:706-800), EdgeGyroRW / EdgeAccRW (G2oTypes.h:633-700), EdgeMono / EdgeStereo
 * (G2oTypes.h:337-437, G2oTypes.cc:352-418; pinhole or KannalaBrandt8, camera index 0/1 of the rig).  Solver: g2o Levenberg-Marquardt
 * with setUserLambdaInit (Optimizer.cc:4884-4896), Schur complement of the landmarks, dense Cholesky of the reduced system.
 * One workgroup runs the whole optimize(iterations) of one window in a single launch; windows are batched. */
typedef struct liba_keyframe {      /* ImuCamPose + the V / G / A vertex estimates of one key frame; row-major 3x3 */
    double Rwb[9], twb[3];          /* body pose (pKF->GetImuRotation / GetImuPosition) */
    double Rcw[2][9], tcw[2][3];    /* per camera, as the ImuCamPose constructor sets them (G2oTypes.cc:43-66); rewritten by every update */
    double v[3], bg[3], ba[3];      /* VertexVelocity, VertexGyroBias, VertexAccBias estimates */
    int32_t pose_fixed;             /* VP->setFixed */
    int32_t has_imu;                /* V / G / A vertices exist (pKFi->bImu) */
    int32_t imu_fixed;              /* V / G / A fixed (the key frame just before the temporal window) */
    int32_t reserved;
} liba_keyframe;                    /* 376 bytes; key frames in ascending mnId order (= g2o's Hessian block order) */
typedef struct liba_rig {           /* calibration members of ImuCamPose (G2oTypes.h:60-72) */
    int32_t n_cams, reserved;
    double Rcb[2][9], tcb[2][3], Rbc[2][9], tbc[2][3];
    double bf;
    int32_t model[2];               /* LBA_CAM_PINHOLE / LBA_CAM_KB8 */
    double p[2][8];                 /* mvParameters widened */
} liba_rig;
typedef struct liba_imu_edge {      /* one IMU::Preintegrated between consecutive key frames: EdgeInertial + EdgeGyroRW + EdgeAccRW */
    int32_t kf1, kf2;               /* previous / current key frame (indices into the window's key frames) */
    float dR[9], dV[3], dP[3];      /* Preintegrated::dR, dV, dP (CV_32F) */
    float JRg[9], JVg[9], JVa[9], JPg[9], JPa[9];
    float b[6];                     /* Preintegrated::b = bax bay baz bwx bwy bwz (the bias the measurements were integrated with) */
    float dT, pad;
    double huber;                   /* 0 = no robust kernel; sqrt(16.92) for i == N-1 or bRecInit (Optimizer.cc:5005-5012) */
    double info[81];                /* EdgeInertial::information() as the reference leaves it (constructor clean-up, x1e-2 for i == N-1) */
    double info_g[9], info_a[9];    /* EdgeGyroRW / EdgeAccRW information (Optimizer.cc:5024-5043) */
} liba_imu_edge;                    /* 1080 bytes */
#define LIBA_MAX_FREE 32            /* optimisable key frames per window (maxOpt is 10, or 25 for bLarge: Optimizer.cc:4758-4764) */
typedef struct liba_problem {
    liba_keyframe* kfs;             /* [batch][cap_kf]  UPDATED IN PLACE */
    const int32_t* n_kf;            /* [batch] */
    const liba_rig* rigs;           /* window b uses rigs[b * rig_stride] (rig_stride 0 = one shared rig) */
    double* points;                 /* [batch][cap_l][3]  UPDATED IN PLACE */
    const int32_t* n_points;        /* [batch] */
    const lba_edge* edges;          /* [batch][cap_e]  landmark-major; pose = key-frame index, kind = LBA_EDGE_MONO / LBA_EDGE_STEREO, cam = camera of the rig */
    const int32_t* n_edges;         /* [batch] */
    const liba_imu_edge* imu;       /* [batch][cap_i]  in the reference's insertion order (newest first) */
    const int32_t* n_imu;           /* [batch] */
    int32_t cap_kf, cap_l, cap_e, cap_i, rig_stride;
    int32_t max_free;               /* upper bound of optimisable key frames in any window (<= LIBA_MAX_FREE): sizes the reduced system */
    double huber_mono, huber_stereo;   /* thHuberMono / thHuberStereo (Optimizer.cc:5071-5073) */
} liba_problem;
size_t liba_workspace_bytes(const liba_problem* prob, int batch);
/* optimizer.optimize(iterations) (Optimizer.cc:5225).  d_stats: [batch][5] device doubles = {iterations run (-1: invalid window), final
 * activeRobustChi2 (err_end), final lambda, lambda trials, initial activeRobustChi2 (err)}.  Asynchronous on `stream`. */
int liba_optimize(const liba_problem* prob, int batch, double lambda_init, int iterations, void* d_workspace, double* d_stats, void* stream);
/* computeActiveErrors: per visual edge chi2 and isDepthPositive (the outlier pass of Optimizer.cc:5237-5275 reads both), per preintegration
 * {EdgeInertial, EdgeGyroRW, EdgeAccRW} chi2, and activeRobustChi2 per window.  Any output may be NULL. */
int liba_compute_errors(const liba_problem* prob, int batch, double* d_vis_chi2, uint8_t* d_vis_depth_pos, double* d_imu_chi2,
                        double* d_robust_sum, void* stream);

/* Optimizer::PoseInertialOptimizationLastKeyFrame(Frame*, bool bRecInit) (reference src/Optimizer.cc:7665-8067): the per-frame optimisation of
 * the inertial tracking modes (Tracking.cc: after TrackLocalMap's matching when the map was updated).  Frame pose / velocity / gyro bias / acc bias free
 * (15 unknowns), the last key frame's vertices fixed; EdgeMonoOnlyPose / EdgeStereoOnlyPose (G2oTypes.h:387-421, 463-491) as pose_edge records
 * (kind = LBA_EDGE_MONO / LBA_EDGE_STEREO, cam = camera of the rig, kind |= LIBA_EDGE_CLOSE iff mTrackDepth < 10), one EdgeInertial + EdgeGyroRW +
 * EdgeAccRW (liba_imu_edge with kf1 = key frame, kf2 = frame; huber ignored).  4 rounds x optimize(10) of g2o's Gauss-Newton with the dense solver,
 * chi2 re-classification {12, 7.5, 5.991, 5.991} / {15.6, 9.8, 7.815, 7.815} with the 1.5x bClose rule, Huber dropped for the last round, the
 * < 30 inliers recovery pass.  One wave per frame, single launch.  Outputs: d_frames updated in place (SetImuPoseVelocity + mImuBias), mvbOutlier per
 * edge, the 15x15 row-major Hessian of the final state (ConstraintPoseImu::H, :8040-8062), and the return value nInitialCorrespondences - nBad. */
#define LIBA_EDGE_CLOSE 0x100
int liba_pose_inertial_kf(liba_keyframe* d_frames, const liba_keyframe* d_keyframes, const liba_rig* d_rigs, int rig_stride, const pose_edge* d_edges,
                          const int32_t* d_n_edges, int cap_e, const liba_imu_edge* d_imu, int batch, int rec_init, uint8_t* d_outlier, double* d_H,
                          int32_t* d_n_good, void* stream);

/* Optimizer::PoseInertialOptimizationLastFrame(Frame*, bool bRecInit) (reference src/Optimizer.cc:8068-8415): as liba_pose_inertial_kf, but the previous
 * FRAME's pose / velocity / biases are free too (30 unknowns) and tied down by EdgePriorPoseImu (G2oTypes.h:737-784; Huber delta 5) built from the
 * ConstraintPoseImu the previous call left (liba_prior: its members after the constructor's eigenvalue clean-up, H row-major); the preintegration is
 * mpImuPreintegratedFrame (kf1 = previous frame, kf2 = frame); chi2Mono = 5.991 in every round.  d_prev_frames is updated in place as well.  d_H =
 * the final 30x30 Hessian marginalised over the previous frame (Optimizer::Marginalize, :5366-5450) = the 15x15 block handed to the new ConstraintPoseImu. */
typedef struct liba_prior {
    double Rwb[9], twb[3], vwb[3], bg[3], ba[3];   /* ConstraintPoseImu linearisation point */
    double H[225];                                 /* ConstraintPoseImu::H, row-major */
} liba_prior;
int liba_pose_inertial_lastframe(liba_keyframe* d_frames, liba_keyframe* d_prev_frames, const liba_rig* d_rigs, int rig_stride, const pose_edge* d_edges,
                                 const int32_t* d_n_edges, int cap_e, const liba_imu_edge* d_imu, const liba_prior* d_priors, int batch, int rec_init,
                                 uint8_t* d_outlier, double* d_H, int32_t* d_n_good, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Device-memory helpers so that adapters written against this header need no HIP headers.
 * ------------------------------------------------------------------------------------------------------- */
int orb_device_count(void);
int orb_dev_alloc(int device, size_t bytes, void** d_ptr);
int orb_dev_free(void* d_ptr);
int orb_host_alloc(size_t bytes, void** h_ptr);   /* page-locked host memory: copies from / to it are real asynchronous DMA (a pageable copy is staged and
                                                     synchronous inside the runtime); the adapters' packed staging blocks live in it */
int orb_host_free(void* h_ptr);
int orb_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes, void* stream);   /* asynchronous on stream */
int orb_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes, void* stream);   /* asynchronous on stream */
int orb_memset(void* d_dst, int value, size_t bytes, void* stream);
int orb_stream_sync(void* stream);   /* NULL = default stream */

#ifdef __cplusplus
}
#endif
#endif /* ORBHIP_H */
