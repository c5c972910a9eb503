#include "conv_dispatch_impl.h"
namespace atdn {
ATDN_INSTANTIATE_CONV(MODE_TAP, EpiGruQ)
ATDN_INSTANTIATE_CONV(MODE_TAP, EpiFlowDelta)
ATDN_INSTANTIATE_CONV(MODE_ROW, EpiBias<ACT_NONE>)
}
