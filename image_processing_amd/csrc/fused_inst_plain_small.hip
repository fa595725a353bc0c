// Instantiations of the fused chain kernel: BGR8 / RGB8 input, range capacity 1 and 2.
#include "fused_kernel.hpp"

namespace ipf {
IPF_DEFINE_LAUNCH_VARIANT(launch_variant_plain_small, false, 1, 2)
}  // namespace ipf
