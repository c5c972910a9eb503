#include "conv_dispatch_impl.h"
namespace atdn {
ATDN_INSTANTIATE_CONV(MODE_ROW, EpiMishBN)
ATDN_INSTANTIATE_CONV(MODE_ROW, EpiMishBNSkipMishBN)
}
