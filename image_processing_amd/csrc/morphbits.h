// Bit-plane morphology back end (morphbits.hip): whole erode/dilate list + moments + final mask in one launch.
#pragma once
#include <vector>

#include "../../include/imgproc.h"
#include "kernels.h"

namespace ipm {

constexpr int MAX_SPANS = 16;

struct Span {          // the element rows that share one horizontal span
    int8_t lx, hx;     // x offsets of the span relative to the anchor
    int16_t dy_lo, dy_hi;  // first / last row offset (relative to the anchor row) that uses it
    uint32_t rows;     // bit (dy - dy_lo): the element row at dy uses this span; 0 = every row of dy_lo..dy_hi (merged RECT)
};

struct Group {         // one (possibly merged) erode/dilate, in the form the kernel walks
    int kind;          // 0 erode, 1 dilate
    int n_spans;
    int t_in, b_in;    // tile rows at the top / bottom that are already invalid before this group (reach of earlier groups)
    int t_out, b_out;  // ... and after it
    Span spans[MAX_SPANS];
};

struct Plan {
    std::vector<Group> groups;
    int hwl = 0, hwr = 0, ht = 0, hb = 0;  // total reach of the list: halo words left/right, rows top/bottom
};

// false when the list cannot run on bit planes (an element row that is not one contiguous span, reach too large)
bool plan_bitmorph(const imgproc_pipeline &pipe, const ipk::Element *elems, Plan *out);

// d_thresh / d_masks: [planes][h][w] bytes (d_masks may be null: moments only); moments added into d_mom[planes][6].
// d_tbits != null: the inRange planes come as bits instead, [planes][h][bit_words(w)] words, bit i of word j = pixel
// 32 j + i (what the front kernel and the inRange kernel write for this back end); d_thresh is then ignored.
inline int bit_words(int w) { return (w + 31) / 32; }
hipError_t launch_bitmorph(const Plan &plan, const Group *d_groups, const ipk::u8 *d_thresh, ipk::u8 *d_masks,
                           unsigned long long *d_mom, int w, int h, int planes, int second_order, hipStream_t stream,
                           const uint32_t *d_tbits = nullptr);

}  // namespace ipm
