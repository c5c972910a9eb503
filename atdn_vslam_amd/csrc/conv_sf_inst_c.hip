#include "conv_sf_dispatch_impl.h"
namespace atdn {
ATDN_INSTANTIATE_CONV_SF(SfQK)
ATDN_INSTANTIATE_CONV_SF(SfVT)
}
