#include "fused.h"

namespace ipf {

bool plan_fused(const imgproc_pipeline &, const ipk::Element *, const ipk::Taps &, const ipk::Taps &,
                const ipk::Bounds &, FusedPlan *)
{
    return false;
}

hipError_t launch_fused(const FusedPlan &, const ipk::u8 *, size_t, size_t, int, int, int, ipk::u8 *,
                        unsigned long long *, const int *, hipStream_t)
{
    return hipErrorNotSupported;
}

}  // namespace ipf
