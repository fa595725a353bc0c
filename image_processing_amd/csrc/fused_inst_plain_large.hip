// Instantiations of the fused chain kernel: BGR8 / RGB8 input, range capacity 4 and 8.
#include "fused_kernel.hpp"

namespace ipf {
IPF_DEFINE_LAUNCH_VARIANT(launch_variant_plain_large, false, 4, 8)
}  // namespace ipf
