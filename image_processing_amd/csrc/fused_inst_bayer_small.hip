// Instantiations of the fused chain kernel: BayerRG8 input, range capacity 1 and 2.
#include "fused_kernel.hpp"

namespace ipf {
IPF_DEFINE_LAUNCH_VARIANT(launch_variant_bayer_small, true, 1, 2)
}  // namespace ipf
