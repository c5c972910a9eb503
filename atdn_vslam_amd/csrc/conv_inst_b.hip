#include "conv_dispatch_impl.h"
namespace atdn {
ATDN_INSTANTIATE_CONV(MODE_TAP, EpiBiasStats)
ATDN_INSTANTIATE_CONV(MODE_ROW, EpiBiasStats)
ATDN_INSTANTIATE_CONV(MODE_TAP, EpiBiasReluAddRelu)
}
