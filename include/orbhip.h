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
/* Copies level to host; border = 0 (plane) or 19 (reference layout incl. reflected frame), out is tightly packed. */
int orbx_copy_level(orbx_handle h, int frame, int level, int border, uint8_t* out);

/* Stage-level taps for parity tests (valid after an extract call; host outputs):
 * FAST candidates of (frame, level) = vToDistributeKeys (ORBextractor.cc:776,845-850) as (x,y,score) int triples
 * in arbitrary order (the set is what the octree consumes); returns count via n_out. */
int orbx_debug_candidates(orbx_handle h, int frame, int level, int32_t* xys, int cap, int* n_out);
/* Octree output of (frame, level) in list order (DistributeOctTree result, ORBextractor.cc:737-761):
 * (x,y,score) int triples, coordinates relative to minBorder like the reference at that point. */
int orbx_debug_selected(orbx_handle h, int frame, int level, int32_t* xys, int cap, int* n_out);

/* Device time of the last batch call's kernels, measured with HIP events on the launch stream:
 * ms[0]=pyramid ms[1]=FAST ms[2]=octree ms[3]=orient+blur+rBRIEF ms[4]=total.  Synchronises the stream. */
int orbx_last_timing(orbx_handle h, float* ms5);

#ifdef __cplusplus
}
#endif
#endif /* ORBHIP_H */
