// Fused chain kernel (fused.hip): BGR/RGB -> HSV -> Gaussian -> inRange -> morphology -> mask + moments
// in one launch over a whole batch, for the pipelines plan_fused() accepts.
#pragma once
#include "../../include/imgproc.h"
#include "kernels.h"

namespace ipf {

struct FusedPlan {
    int variant;      // kernel specialisation chosen by plan_fused
    int bidx;         // 0 BGR input, 2 RGB input
    int second_order; // also accumulate m20/m11/m02
    int bayer;        // input is BayerRG8 (1 B/px): the kernel demosaics while loading (two halo lanes per side)
    int pre_k;        // pre-HSV blur ksize on the raw channels (1 = none, 3)
    int post_k;       // post-HSV blur ksize (1 = none)
    int mode;         // 0: whole chain in the kernel; 1: front end only (inRange bytes out; morphbits.hip finishes)
    int n_groups;     // merged morphology groups
    int kind[8];      // per group: 0 erode, 1 dilate
    int lo_x[8], hi_x[8], lo_y[8], hi_y[8];  // per group: window offset range [lo, hi] (rect, merged)
    ipk::Bounds bounds;
};

// true when the pipeline can run fused; fills *out
bool plan_fused(const imgproc_pipeline &pipe, const ipk::Element *elems, const ipk::Taps &pre, const ipk::Taps &post,
                const ipk::Bounds &bounds, FusedPlan *out);

// front end for pipelines whose morphology list does not fit the fused kernel: [blur3] HSV [blur5] inRange -> bytes
bool plan_front(const imgproc_pipeline &pipe, const ipk::Bounds &bounds, FusedPlan *out);

// launch-time conditions (alignment, width % 4) the fused kernel needs; false -> run stage by stage
bool can_launch_fused(const FusedPlan &plan, const void *d_frames, size_t frame_stride, size_t row_stride, int w, int h,
                      const void *d_masks);

// frames: n frames at d_frames + i*frame_stride; masks (may be null): [n][n_ranges][h][w]; moments added into
// d_mom[n][n_ranges][6] (caller zeroes)
// overlapped: another launch of the context is in flight on the other batch stream (imgproc_batch_enqueue), so the end of
// this launch is covered by that one and segments can be tall (fewer look-ahead rows); alone, ~100-row segments win.
hipError_t launch_fused(const FusedPlan &plan, const ipk::u8 *d_frames, size_t frame_stride, size_t row_stride, int w,
                        int h, int n, ipk::u8 *d_masks, unsigned long long *d_mom, const int *d_tables,
                        hipStream_t stream, uint32_t *d_bits = nullptr, bool overlapped = false);
// A front plan (plan_front) can hand its inRange result to the bit-plane back end as bits instead of bytes when the
// strips' borders fall on byte borders of the bitmap: pass d_bits ([n][n_ranges][h][(w+31)/32] words) instead of d_masks.
inline bool front_writes_bits(int w) { return (w & 3) == 0; }

}  // namespace ipf
