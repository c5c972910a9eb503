#include "conv_sf_dispatch_impl.h"
namespace atdn {
#ifdef ATDN_CONV_STAMP   // diagnostic build: block-lifetime stamps of the ConvGRU kernels (conv_sf6.h, tools/diag/conv_stamps.py)
__device__ unsigned long long atdn_conv_stamps_dev[4][ATDN_CONV_STAMP_SLOTS][12];
template <> struct conv_stamp_kind<SfGruZR> : std::integral_constant<int, 0> {};
template <> struct conv_stamp_kind<SfGruQ> : std::integral_constant<int, 1> {};
#endif
ATDN_INSTANTIATE_CONV_SF(SfGruZR)
ATDN_INSTANTIATE_CONV_SF(SfGruQ)
}
#ifdef ATDN_CONV_STAMP
extern "C" int atdn_conv_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(atdn::atdn_conv_stamps_dev), sizeof(unsigned long long) * 4 * ATDN_CONV_STAMP_SLOTS * 12);
}
#endif
