#include "conv_dispatch_impl.h"
namespace atdn {
ATDN_INSTANTIATE_CONV(MODE_TAP, EpiStoreT)
ATDN_INSTANTIATE_CONV(MODE_TAP, EpiAggregate)
ATDN_INSTANTIATE_CONV(MODE_TAP, EpiGruZR)
}
