// Bit-plane morphology back end (morphbits.hip): whole erode/dilate list + moments + final mask in one launch.
#pragma once
#include <vector>

#include "../../include/imgproc.h"
#include "kernels.h"

namespace ipm {

struct Group {      // one (possibly merged) erode/dilate
    int kind;       // 0 erode, 1 dilate
    int nrows, ay;  // element rows and anchor row (a merged RECT group may be taller than 31)
    int rect;       // all rows share the span lx[0]..hx[0]: runs separably
    int8_t lx[32], hx[32];  // per element row: x offsets of its span relative to the anchor (lx > hx: empty row)
};

struct Plan {
    std::vector<Group> groups;
    int hwl = 0, hwr = 0, ht = 0, hb = 0;  // total reach of the list: halo words left/right, rows top/bottom
};

// false when the list cannot run on bit planes (an element row that is not one contiguous span, reach too large)
bool plan_bitmorph(const imgproc_pipeline &pipe, const ipk::Element *elems, Plan *out);

// d_thresh / d_masks: [planes][h][w] bytes (d_masks may be null: moments only); moments added into d_mom[planes][6]
hipError_t launch_bitmorph(const Plan &plan, const Group *d_groups, const ipk::u8 *d_thresh, ipk::u8 *d_masks,
                           unsigned long long *d_mom, int w, int h, int planes, int second_order, hipStream_t stream);

}  // namespace ipm
