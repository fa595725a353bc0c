// Instantiations of the fused chain kernel: BayerRG8 input, range capacity 4 and 8.
#include "fused_kernel.hpp"

namespace ipf {
IPF_DEFINE_LAUNCH_VARIANT(launch_variant_bayer_large, true, 4, 8)
}  // namespace ipf
