/*
 * imgproc.h — C ABI of libimgproc_hip, the MI355X (gfx950) replacement for the OpenCV CPU chain
 * that the image_processing ROS nodes run in their camera callback.
 *
 * Reference call sites this library replaces (paths relative to the reference repo):
 *   image_processing/src/objectPolygonPath.cpp:182-190  GaussianBlur 3x3, cvtColor BGR2HSV,
 *        GaussianBlur 5x5, inRange, morphOps, GRAY2BGR + bitwise_and       (polygon_path node)
 *   image_processing/src/objectPolygonPath.cpp:414-433  morphOps (E3x4, D8x2, E3x2, D8x2, E3x4)
 *   image_processing/src/pose_estimate.cpp:203-206      cvtColor BGR2HSV, inRange (+ commented
 *        GaussianBlur 5x5 and morphOps)                                     (pose_estimate node)
 *   image_processing/src/pose_estimate.cpp:405-415      morphOps (E3x2, D8x2, E3x2)
 *   image_processing/src/objectPolygonPath.cpp:449-452  moments(...).m00 area gate / centroid
 *   image_processing/src/pose_estimate.cpp:346-354      dynamicReconfigureCallback (6 ints 0..256)
 *
 * Plain C, plain pointers and sizes.  No C++ exception crosses this boundary; every function
 * returns IMGPROC_OK (0) or a negative imgproc_status.  A context is single-caller (it mirrors the
 * node's single spinner thread, pose_estimate.cpp:394-399); different contexts are independent.
 * There is NO CPU fallback: without a HIP device every entry point that computes fails with
 * IMGPROC_ENODEV / IMGPROC_EHIP.
 */
#ifndef IMGPROC_H
#define IMGPROC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IMGPROC_VERSION 0x000100 /* 0.1.0 */
#define IMGPROC_MAX_RANGES 8
#define IMGPROC_MAX_MORPH 32
#define IMGPROC_MAX_KSIZE 31
#define IMGPROC_MAX_TICKETS 4 /* device batches in flight per context (imgproc_batch_enqueue) */

typedef struct imgproc_ctx imgproc_ctx;

enum imgproc_status {
    IMGPROC_OK = 0,
    IMGPROC_EINVAL = -1,       /* bad argument */
    IMGPROC_ENOMEM = -2,       /* host or device allocation failed */
    IMGPROC_EHIP = -3,         /* a HIP runtime call failed (see imgproc_last_error) */
    IMGPROC_ESTATE = -4,       /* call order (wait without submit, stage not kept, ...) */
    IMGPROC_ESIZE = -5,        /* frame larger than the context was created for */
    IMGPROC_ENODEV = -6,       /* no usable HIP device */
    IMGPROC_EUNSUPPORTED = -7  /* valid request this build cannot serve */
};

/* sensor_msgs encodings the node may hand over unconverted (replaces cv_bridge::toCvCopy(msg,BGR8),
 * pose_estimate.cpp:180 / objectPolygonPath.cpp:173; the camera publishes rgb8,
 * pylon_camera/src/custom_pylon_node.cpp:157). */
enum imgproc_encoding { IMGPROC_ENC_BGR8 = 0, IMGPROC_ENC_RGB8 = 1, IMGPROC_ENC_BAYER_RGGB8 = 2 };

enum imgproc_morph_kind { IMGPROC_ERODE = 0, IMGPROC_DILATE = 1 };

/* values match cv::MorphShapes so getStructuringElement(MORPH_RECT, ...) maps 1:1
 * (objectPolygonPath.cpp:31-32, pose_estimate.cpp:40-41) */
enum imgproc_shape { IMGPROC_RECT = 0, IMGPROC_CROSS = 1, IMGPROC_ELLIPSE = 2, IMGPROC_CUSTOM = 3 };

/* stage outputs that imgproc_fetch_stage can return (MASK always; the others are kept at submit when
 * pipeline.keep_stages != 0, otherwise rebuilt from the slot's raw frame when they are asked for) */
enum imgproc_stage {
    IMGPROC_STAGE_BGR = 0,    /* u8x3: after demosaic / channel order normalisation / pre-blur   */
    IMGPROC_STAGE_HSV = 1,    /* u8x3: cvtColor BGR2HSV output                                   */
    IMGPROC_STAGE_BLUR = 2,   /* u8x3: HSV after the post-HSV GaussianBlur (== HSV when k<=1)     */
    IMGPROC_STAGE_THRESH = 3, /* u8x1: inRange output of range `range`                            */
    IMGPROC_STAGE_MASK = 4,   /* u8x1: after the morphology list (the mask moments are taken of)  */
    IMGPROC_STAGE_MASKED = 5  /* u8x3: frame & GRAY2BGR(mask)  (objectPolygonPath.cpp:189-190)    */
};

typedef struct imgproc_config {
    int device;     /* HIP device ordinal */
    int max_width;  /* largest frame the context will see */
    int max_height;
    int n_slots;    /* in-flight frames for submit/wait (>=1; 2 = double buffering) */
    int max_batch;  /* largest n for imgproc_process_batch_device (0 -> 1) */
} imgproc_config;

/* One erode/dilate call: erode(image,image,erodeElement) etc. (objectPolygonPath.cpp:416-432).
 * anchor (-1,-1) = element centre (kw/2, kh/2), OpenCV's default.  `custom` (kh*kw bytes,
 * non-zero = member) is read only when shape == IMGPROC_CUSTOM and is copied by the library. */
typedef struct imgproc_morph {
    int kind;  /* imgproc_morph_kind */
    int shape; /* imgproc_shape */
    int kw, kh;
    int anchor_x, anchor_y;
    const uint8_t *custom;
} imgproc_morph;

/* The chain, in the reference's call order (objectPolygonPath.cpp:182-188):
 * [demosaic] -> [GaussianBlur pre_blur_k] -> BGR2HSV -> [GaussianBlur post_blur_k] -> inRange (per
 * range) -> morph[0..n_morph) -> moments.  ksize 0 or 1 = no blur; sigma is always 0 as in the
 * reference. */
typedef struct imgproc_pipeline {
    int encoding; /* imgproc_encoding of the frames handed to submit / process_batch */
    int pre_blur_k;
    int post_blur_k;
    int n_morph;
    imgproc_morph morph[IMGPROC_MAX_MORPH];
    int keep_stages;   /* produce and keep every stage output at submit (stage-by-stage path); 0: fast path, stage
                          images other than MASK are rebuilt on demand inside imgproc_fetch_stage */
    int force_unfused; /* run stage-by-stage kernels even when a fused kernel exists (parity tests) */
    int second_order;  /* also compute m20/m11/m02 and theta; 0: they are reported as 0 */
} imgproc_pipeline;

/* Raster moments of the final mask (== cv::moments(mask, true)); all sums are exact integers.
 * found = (m00 != 0); cx = m10/m00, cy = m01/m00 in double, or -1 when !found.
 * theta = 0.5*atan2(2*mu11, mu20-mu02) (principal axis, radians), 0 when !found. */
typedef struct imgproc_moments {
    uint64_t m00, m10, m01, m20, m11, m02;
    double cx, cy, theta;
    int found;
    int reserved;
} imgproc_moments;

typedef struct imgproc_result {
    int n_ranges;
    int reserved;
    imgproc_moments r[IMGPROC_MAX_RANGES];
} imgproc_result;

/* ---- lifecycle -------------------------------------------------------------------------------- */
int imgproc_version(void);
const char *imgproc_strerror(int status);
/* text of the last failure on this context (HIP error string etc.); never NULL */
const char *imgproc_last_error(const imgproc_ctx *ctx);
int imgproc_device_count(int *count);
int imgproc_create(imgproc_ctx **out, const imgproc_config *cfg);
void imgproc_destroy(imgproc_ctx *ctx);

/* ---- configuration (replaces the globals the node's callbacks write) -------------------------- */
int imgproc_set_pipeline(imgproc_ctx *ctx, const imgproc_pipeline *p);
/* Raw ints exactly as dynamic_reconfigure delivers them (cfg/hsvThresh.cfg:8-13, range 0..256).
 * The library applies cv::inRange's rules: lo>hi, lo>255 or hi<0 on a channel => never matches;
 * otherwise bounds clamp to [0,255].  The node's H_MIN/H_MAX swap (pose_estimate.cpp:348-349)
 * stays in the node shim. */
int imgproc_set_ranges(imgproc_ctx *ctx, int n, const int32_t (*lo)[3], const int32_t (*hi)[3]);

/* ---- streaming path: what imageCb calls per frame (pose_estimate.cpp:175, objectPolygonPath.cpp:168)
 * submit copies `data` into the slot's pinned staging buffer (the caller may free it on return),
 * uploads on the slot's own stream and enqueues the chain behind it; wait blocks until that slot's
 * results are on the host and reports them for the ranges the frame was submitted with. */
int imgproc_submit(imgproc_ctx *ctx, int slot, const uint8_t *data, size_t stride, int w, int h);
/* submit without the staging copy, for callers whose frames already sit in page-locked host memory (a camera
 * driver's DMA ring, imgproc_host_alloc): the upload reads `data` itself, so it must stay untouched until
 * imgproc_wait(slot) returns.  No counterpart in the reference (its frames arrive in a std::vector); with pageable
 * memory the call still works but the runtime stages the copy itself and the call may block. */
int imgproc_submit_pinned(imgproc_ctx *ctx, int slot, const uint8_t *data, size_t stride, int w, int h);
int imgproc_wait(imgproc_ctx *ctx, int slot, imgproc_result *out);
int imgproc_fetch_stage(imgproc_ctx *ctx, int slot, int stage, int range, uint8_t *dst, size_t dst_stride);

/* ---- batch paths -------------------------------------------------------------------------------
 * Host frames: frames[i] starts at frames + i*frame_stride; pipelined through the slots. */
int imgproc_process_batch(imgproc_ctx *ctx, const uint8_t *frames, size_t frame_stride, size_t row_stride,
                          int w, int h, int n, imgproc_result *out);
/* Device-resident frames (the roofline configuration): d_frames is a device pointer to n frames,
 * d_masks (may be NULL) receives [n][n_ranges][h][w] final masks, out (host) receives n results. */
int imgproc_process_batch_device(imgproc_ctx *ctx, const void *d_frames, size_t frame_stride, size_t row_stride,
                                 int w, int h, int n, void *d_masks, imgproc_result *out);
/* The same in two halves, so that several device batches can be in flight (offline bag processing, the benchmark):
 * enqueue puts batch `ticket` (0 .. IMGPROC_MAX_TICKETS - 1) on a stream of the context and returns; wait blocks until
 * it is done and fills out[n] (out may be NULL).  d_frames / d_masks must stay valid until the wait.  Tickets are
 * independent of each other: pipelines that run as one or two launches per batch put every ticket on a stream of its
 * own, so the kernels of batches in flight overlap on the GPU (two in flight: 3-16 % more frames/s than one stream,
 * DESIGN.md 4; three help the two-launch chains a little more, four are slower; IMGPROC_BATCH_STREAMS=1 in the
 * environment at imgproc_create keeps all tickets on one stream) -- the d_masks of batches in flight must therefore
 * not overlap, and no batch may read what another writes.  process_batch_device is enqueue(ticket 0) + wait(0).
 * Like it, these have no call site in the reference (its nodes see one frame at a time). */
int imgproc_batch_enqueue(imgproc_ctx *ctx, const void *d_frames, size_t frame_stride, size_t row_stride, int w, int h,
                          int n, void *d_masks, int ticket);
int imgproc_batch_wait(imgproc_ctx *ctx, int ticket, imgproc_result *out);

/* imgproc_process_batch_device for callers that cannot allocate (tight serving loops, harnesses that forbid
 * hipMalloc after start-up, processes that share a GPU whose memory is fully committed).  It has no counterpart in
 * the reference.  Same arguments and results as imgproc_process_batch_device, but the library's scratch (the
 * moment sums; the inRange planes between the front kernel and the bit-plane kernel) lives in a static per-GPU
 * device array of 32 MiB + 384 KiB instead of allocated buffers, and the call makes no hipMalloc / hipFree /
 * hipHostMalloc.  Limits: n <= max_batch, n * n_ranges <= 8192, and for pipelines that run as front kernel +
 * bit-plane morphology the inRange planes (n * n_ranges * h * 4 * ceil(w / 32) bytes as bit planes;
 * n * n_ranges * w * h when w % 4 != 0) <= 32 MiB (IMGPROC_ESIZE beyond); pipelines or frame shapes that run stage
 * by stage return IMGPROC_EUNSUPPORTED.  Synchronous on the context's own stream: it enqueues there, waits for the
 * stream and returns with `out` filled, so it cannot be part of a caller's HIP graph capture (it does not take the
 * caller's stream, and the wait is illegal under capture).  Every call on a GPU shares that GPU's array: while one call
 * is inside it, a second one on the same GPU (another context, another thread) returns IMGPROC_ESTATE. */
int imgproc_process_batch_device_noalloc(imgproc_ctx *ctx, const void *d_frames, size_t frame_stride,
                                         size_t row_stride, int w, int h, int n, void *d_masks, imgproc_result *out);

/* ---- single stages on host buffers: one entry point per cv:: call on the path ----------------- */
/* cvtColor(src, dst, CV_BGR2HSV)  objectPolygonPath.cpp:183, pose_estimate.cpp:203; encoding BGR8|RGB8 */
int imgproc_cvt_bgr2hsv(imgproc_ctx *ctx, const uint8_t *src, size_t sstride, uint8_t *dst, size_t dstride,
                        int w, int h, int encoding);
/* GaussianBlur(src, dst, Size(k,k), 0, 0)  objectPolygonPath.cpp:182,184; channels 1 or 3 */
int imgproc_gaussian_blur(imgproc_ctx *ctx, const uint8_t *src, size_t sstride, uint8_t *dst, size_t dstride,
                          int w, int h, int channels, int ksize);
/* inRange(src, Scalar lo, Scalar hi, dst)  objectPolygonPath.cpp:186, pose_estimate.cpp:205 */
int imgproc_in_range(imgproc_ctx *ctx, const uint8_t *src, size_t sstride, uint8_t *dst, size_t dstride,
                     int w, int h, const int32_t lo[3], const int32_t hi[3]);
/* erode / dilate (src, dst, element)  objectPolygonPath.cpp:416-432, pose_estimate.cpp:407-414 */
int imgproc_morphology(imgproc_ctx *ctx, const uint8_t *src, size_t sstride, uint8_t *dst, size_t dstride,
                       int w, int h, const imgproc_morph *op);
/* moments(mask, true)  objectPolygonPath.cpp:449-450 */
int imgproc_mask_moments(imgproc_ctx *ctx, const uint8_t *mask, size_t stride, int w, int h, imgproc_moments *out);
/* cv_bridge bayer_rggb8 -> bgr8 (cv::COLOR_BayerBG2BGR) at the toCvCopy step, pose_estimate.cpp:180 */
int imgproc_demosaic_rggb(imgproc_ctx *ctx, const uint8_t *src, size_t sstride, uint8_t *dst, size_t dstride,
                          int w, int h);
/* cvtColor(GRAY2BGR) + bitwise_and(frame, maskFrame, procFrame)  objectPolygonPath.cpp:189-190 */
int imgproc_mask_and(imgproc_ctx *ctx, const uint8_t *bgr, size_t sstride, const uint8_t *mask, size_t mstride,
                     uint8_t *dst, size_t dstride, int w, int h);
/* getStructuringElement(shape, Size(kw,kh))  objectPolygonPath.cpp:31-32 (host only, no device) */
int imgproc_structuring_element(int shape, int kw, int kh, uint8_t *elem);
/* 8.8 fixed-point taps GaussianBlur(...,Size(k,k),0,0) uses for CV_8U (host only) */
int imgproc_gaussian_taps(int ksize, int32_t *taps);

/* ---- device memory helpers for callers without their own HIP allocator ------------------------ */
int imgproc_dev_alloc(imgproc_ctx *ctx, size_t bytes, void **dptr);
int imgproc_dev_free(imgproc_ctx *ctx, void *dptr);
int imgproc_dev_upload(imgproc_ctx *ctx, void *dptr, const void *src, size_t bytes);
/* page-locked host memory for imgproc_submit_pinned */
int imgproc_host_alloc(imgproc_ctx *ctx, size_t bytes, void **hptr);
int imgproc_host_free(imgproc_ctx *ctx, void *hptr);
int imgproc_dev_download(imgproc_ctx *ctx, void *dst, const void *dptr, size_t bytes);

/* ---- measurement ------------------------------------------------------------------------------
 * With profiling on, every device batch brackets its chain kernels with HIP events on the
 * library's own stream; last_kernel_ms returns the device time of the chain kernel(s) of the batch
 * that was waited for last, and how many launches that was. */
int imgproc_set_profiling(imgproc_ctx *ctx, int on);
int imgproc_last_kernel_ms(imgproc_ctx *ctx, float *ms, int *launches);
/* The same two events as times on the device clock, in ms since the last imgproc_set_profiling(ctx, 1): with two
 * batches in flight on two streams the spans of consecutive batches overlap, and a caller that wants the rate of the
 * kernels while they run needs the union of the spans, not the sum of their lengths (bench.py: roofline). */
int imgproc_last_kernel_span(imgproc_ctx *ctx, float *t0_ms, float *t1_ms);
/* 1 if the current pipeline runs as one fused kernel, 0 if stage-by-stage */
int imgproc_pipeline_is_fused(const imgproc_ctx *ctx);
/* human-readable description of how the current pipeline will run (which kernels), for logs and tests */
int imgproc_pipeline_plan(const imgproc_ctx *ctx, char *buf, size_t len);
int imgproc_synchronize(imgproc_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* IMGPROC_H */
