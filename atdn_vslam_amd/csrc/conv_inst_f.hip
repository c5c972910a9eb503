#include "conv_dispatch_impl.h"
#include "epilogues_sf.h"
namespace atdn {
ATDN_INSTANTIATE_CONV(MODE_ROW, EpiMishBN)
ATDN_INSTANTIATE_CONV(MODE_ROW, EpiMishBNSkipMishBN)
ATDN_INSTANTIATE_CONV(MODE_ROW, SfBias<ACT_RELU>)
}
