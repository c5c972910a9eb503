#include "conv_sf_dispatch_impl.h"
namespace atdn {
ATDN_INSTANTIATE_CONV_SF(SfBias<ACT_NONE>)
ATDN_INSTANTIATE_CONV_SF(SfBias<ACT_RELU>)
ATDN_INSTANTIATE_CONV_SF(EpiBias<ACT_NONE>)
}
