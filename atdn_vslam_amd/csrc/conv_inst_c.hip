#include "conv_dispatch_impl.h"
namespace atdn {
ATDN_INSTANTIATE_CONV(MODE_TAP, EpiContextSplit)
ATDN_INSTANTIATE_CONV(MODE_TAP, EpiScale)
ATDN_INSTANTIATE_CONV(MODE_TAP, EpiQK)
}
