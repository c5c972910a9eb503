#include "conv_sf_dispatch_impl.h"
namespace atdn {
ATDN_INSTANTIATE_CONV_SF(EpiBiasStats)
ATDN_INSTANTIATE_CONV_SF(SfBiasReluAddRelu)
ATDN_INSTANTIATE_CONV_SF(SfContextSplit)
}
