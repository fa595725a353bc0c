// ROS-free mirror of the two vision nodes' camera callbacks, calling libimgproc_hip through its C ABI
// (SURVEY.md §8 f-4).  It holds exactly the host logic that sits around the pixel chain in the reference, so that
// the library can be exercised the way the nodes would drive it without ROS / OpenCV in the build:
//   PoseEstimateShim  <- image_processing/src/pose_estimate.cpp      objectTracker::imageCb :175-279, roiCb :289-321,
//                        objectRoiCb :281-287, dynamicReconfigureCallback :346-354 (incl. its H_MIN/H_MAX swap)
//   PolygonPathShim   <- image_processing/src/objectPolygonPath.cpp  objectTracker::imageCb :168-203, morphOps :414-433
// Out of scope and therefore absent: the boosting tracker, the Kalman filter, contours/PCA, all drawing and publishing.
// The tracker ROI is whatever objectRoiCb last set (the reference re-initialises its tracker there).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/imgproc.h"

namespace imgproc_node {

struct HsvThreshConfig {  // cfg/hsvThresh.cfg:8-13 (defaults min 0 / max 256)
    int H_MIN = 0, S_MIN = 0, V_MIN = 0, H_MAX = 256, S_MAX = 256, V_MAX = 256;
};
struct RegionOfInterest {  // sensor_msgs/RegionOfInterest
    uint32_t x_offset = 0, y_offset = 0, height = 0, width = 0;
};

inline void check(imgproc_ctx *c, int rc, const char *what)
{
    if (rc != IMGPROC_OK) throw std::runtime_error(std::string(what) + ": " + imgproc_strerror(rc) + " / " + imgproc_last_error(c));
}

class PoseEstimateShim {
public:
    static constexpr int FRAME_WIDTH = 1920, FRAME_HEIGHT = 1200;  // pose_estimate.cpp:43-44
    // globals of the node (pose_estimate.cpp:53-65)
    int H_MIN = 0, H_MAX = 256, S_MIN = 0, S_MAX = 256, V_MIN = 0, V_MAX = 256;
    double x_obj = 0, y_obj = 0, scale = 1;
    int offsetX = 0, offsetY = 0;
    bool isObjectFound = false;
    double roi_x = 0, roi_y = 0, roi_w = 0, roi_h = 0;  // Rect2d roi (:82)
    // what the callback would publish on <ns>/positionDist (:258-259) plus the raw centroid
    struct Out {
        bool published = false;
        double position_x = 0, position_y = 0;  // metres
        int center_x = 0, center_y = 0;         // tempPoint_ (full-frame pixel, int-truncated like :459,:468)
        uint64_t m00 = 0;
        std::vector<uint8_t> mask;  // pcaFrame after inRange (:205), roi_w x roi_h
        int mask_w = 0, mask_h = 0;
    };

    // enable_blur_and_morph re-enables the node's commented-out lines :204 / :206.  The library then blurs the crop as an
    // image of its own (BORDER_REFLECT_101 at the crop's edge); the reference's GaussianBlur on the submatrix
    // Mat(pcaFrame, roi_) -- HSV inside the ROI after the in-place cvtColor, still BGR outside, no BORDER_ISOLATED --
    // would read the parent's bytes there, so within the blur / morphology reach of the crop border this optional path
    // is not what un-commenting the lines would compute.  The live chain (:203 + :205) is exact.
    explicit PoseEstimateShim(int device = 0, bool enable_blur_and_morph = false)
    {
        imgproc_config cfg = {device, FRAME_WIDTH, FRAME_HEIGHT, 2, 1};
        check(nullptr, imgproc_create(&ctx_, &cfg), "imgproc_create");
        imgproc_pipeline p;
        std::memset(&p, 0, sizeof p);
        p.encoding = IMGPROC_ENC_BGR8;
        if (enable_blur_and_morph) {  // the two commented-out lines :204 and :206
            p.post_blur_k = 5;
            const int seq[6] = {0, 0, 1, 1, 0, 0};  // morphOps :407-414
            p.n_morph = 6;
            for (int i = 0; i < 6; i++) {
                p.morph[i].kind = seq[i] ? IMGPROC_DILATE : IMGPROC_ERODE;
                p.morph[i].shape = IMGPROC_RECT;                  // :40-41
                p.morph[i].kw = p.morph[i].kh = seq[i] ? 8 : 3;
                p.morph[i].anchor_x = p.morph[i].anchor_y = -1;
            }
        }
        check(ctx_, imgproc_set_pipeline(ctx_, &p), "imgproc_set_pipeline");
    }
    ~PoseEstimateShim() { imgproc_destroy(ctx_); }
    PoseEstimateShim(const PoseEstimateShim &) = delete;
    PoseEstimateShim &operator=(const PoseEstimateShim &) = delete;

    void dynamicReconfigureCallback(const HsvThreshConfig &config)  // :346-354, swap kept on purpose
    {
        H_MAX = config.H_MIN;
        H_MIN = config.H_MAX;
        S_MAX = config.S_MAX;
        S_MIN = config.S_MIN;
        V_MAX = config.V_MAX;
        V_MIN = config.V_MIN;
    }
    void objectRoiCb(const RegionOfInterest &m)  // :281-287
    {
        roi_x = m.x_offset; roi_y = m.y_offset; roi_w = m.width; roi_h = m.height;
        isObjectFound = true;
    }
    void roiCb(const RegionOfInterest &m)  // :289-321
    {
        int64_t modOffsetX = (int64_t)(m.x_offset % 32);
        if (modOffsetX > 32 / 2) modOffsetX -= 32;
        int64_t modOffsetY = (int64_t)(m.y_offset % 2);
        if (modOffsetY > 2 / 2) modOffsetY -= 2;
        offsetX = (int)(m.x_offset - modOffsetX);
        offsetY = (int)(m.y_offset - modOffsetY);
    }

    // frame: BGR8, `step` bytes per row (what cv_ptr->image holds after toCvCopy(msg, BGR8), :180-190)
    Out imageCb(const uint8_t *frame, size_t step, int cols, int rows)
    {
        Out o;
        if (!isObjectFound) return o;  // :193
        // roi_ = roi; roi_.x = roi.x - 7.5; ... (:197-201): Rect has int members, the doubles truncate toward zero
        const int rx = (int)(roi_x - 7.5), ry = (int)(roi_y - 7.5), rw = (int)(roi_w + 15), rh = (int)(roi_h + 15);
        if (rx < 0 || ry < 0 || rw <= 0 || rh <= 0 || rx + rw > cols || ry + rh > rows)
            throw std::out_of_range("tracker ROI outside the frame (cv::Mat(pcaFrame, roi_) would assert, :202)");
        const int32_t lo[1][3] = {{H_MIN, S_MIN, V_MIN}}, hi[1][3] = {{H_MAX, S_MAX, V_MAX}};  // :205
        check(ctx_, imgproc_set_ranges(ctx_, 1, lo, hi), "imgproc_set_ranges");
        const uint8_t *crop = frame + (size_t)ry * step + (size_t)rx * 3;
        imgproc_result res;
        check(ctx_, imgproc_submit(ctx_, 0, crop, step, rw, rh), "imgproc_submit");  // cvtColor + inRange (:203-205)
        check(ctx_, imgproc_wait(ctx_, 0, &res), "imgproc_wait");
        o.mask.resize((size_t)rw * rh);
        o.mask_w = rw;
        o.mask_h = rh;
        check(ctx_, imgproc_fetch_stage(ctx_, 0, IMGPROC_STAGE_MASK, 0, o.mask.data(), (size_t)rw), "imgproc_fetch_stage");
        o.m00 = res.r[0].m00;
        if (res.r[0].found) {
            // replaces findContours + PCA mean (:207-221, :449-474) by the raster centroid; int-truncated like :459
            o.center_x = (int)res.r[0].cx + rx;
            o.center_y = (int)res.r[0].cy + ry;
        }
        x_obj = (double)(-1920.0 / 2.0 + (offsetX + o.center_x));  // :226
        y_obj = (double)(-1200.0 / 2.0 + (offsetY + o.center_y));  // :227
        o.position_x = x_obj / scale * 0.07758621 * 1e-3;          // :258
        o.position_y = y_obj / scale * 0.07758621 * 1e-3;          // :259
        o.published = true;
        return o;
    }

private:
    imgproc_ctx *ctx_ = nullptr;
};

class PolygonPathShim {
public:
    static constexpr int FRAME_WIDTH = 1920, FRAME_HEIGHT = 1200;  // objectPolygonPath.cpp:36-37
    int H_MIN = 147, H_MAX = 256, S_MIN = 0, S_MAX = 256, V_MIN = 0, V_MAX = 256;  // trackbar globals :57-62
    struct Out {
        std::vector<uint8_t> thresh;  // hsvImageThresh after morphOps (:188)
        std::vector<uint8_t> proc;    // procFrame = frame & GRAY2BGR(mask) (:189-190)
        imgproc_moments moments;      // raster moments of the mask (the m00 area gate of :449-452)
    };

    explicit PolygonPathShim(int device = 0)
    {
        imgproc_config cfg = {device, FRAME_WIDTH, FRAME_HEIGHT, 2, 1};
        check(nullptr, imgproc_create(&ctx_, &cfg), "imgproc_create");
        imgproc_pipeline p;
        std::memset(&p, 0, sizeof p);
        p.encoding = IMGPROC_ENC_BGR8;
        p.pre_blur_k = 3;   // :182
        p.post_blur_k = 5;  // :184
        const int seq[14] = {0, 0, 0, 0, 1, 1, 0, 0, 1, 1, 0, 0, 0, 0};  // :416-432
        p.n_morph = 14;
        for (int i = 0; i < 14; i++) {
            p.morph[i].kind = seq[i] ? IMGPROC_DILATE : IMGPROC_ERODE;
            p.morph[i].shape = IMGPROC_RECT;  // :31-32
            p.morph[i].kw = p.morph[i].kh = seq[i] ? 8 : 3;
            p.morph[i].anchor_x = p.morph[i].anchor_y = -1;
        }
        p.keep_stages = 0;  // fast path (front kernel + bit planes); procFrame is rebuilt on demand by fetch_stage
        check(ctx_, imgproc_set_pipeline(ctx_, &p), "imgproc_set_pipeline");
    }
    ~PolygonPathShim() { imgproc_destroy(ctx_); }
    PolygonPathShim(const PolygonPathShim &) = delete;
    PolygonPathShim &operator=(const PolygonPathShim &) = delete;

    Out imageCb(const uint8_t *frame, size_t step, int cols, int rows)  // :168-203 (hot path)
    {
        Out o;
        const int32_t lo[1][3] = {{H_MIN, S_MIN, V_MIN}}, hi[1][3] = {{H_MAX, S_MAX, V_MAX}};  // :186
        check(ctx_, imgproc_set_ranges(ctx_, 1, lo, hi), "imgproc_set_ranges");
        imgproc_result res;
        check(ctx_, imgproc_submit(ctx_, 0, frame, step, cols, rows), "imgproc_submit");
        check(ctx_, imgproc_wait(ctx_, 0, &res), "imgproc_wait");
        o.thresh.resize((size_t)cols * rows);
        o.proc.resize((size_t)cols * rows * 3);
        check(ctx_, imgproc_fetch_stage(ctx_, 0, IMGPROC_STAGE_MASK, 0, o.thresh.data(), (size_t)cols), "fetch mask");
        check(ctx_, imgproc_fetch_stage(ctx_, 0, IMGPROC_STAGE_MASKED, 0, o.proc.data(), (size_t)cols * 3), "fetch masked");
        o.moments = res.r[0];
        return o;
    }

private:
    imgproc_ctx *ctx_ = nullptr;
};

}  // namespace imgproc_node
