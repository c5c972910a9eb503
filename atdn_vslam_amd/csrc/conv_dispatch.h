// Tile-shape selection for the implicit-GEMM engine. Definitions are explicitly
// instantiated in conv_inst_*.hip so the (epilogue x tile) kernel family compiles in parallel.
#pragma once
#include "conv_mfma.h"
#include "epilogues.h"

namespace atdn {

struct TileChoice {
  int BM, BN;
  int groups_per_img = 0;  // InstanceNorm statistics groups per image written by a kStats epilogue
  bool counted = false;    // true: groups are 2-D tile quarters and part_cnt holds their sizes
};

// Deterministic: depends only on the problem shape (graph capture + InstanceNorm group bookkeeping rely on it).
inline TileChoice choose_tile(int nimg, int HoWo, int N) {
  if (N <= 32) return {128, 32};
  if (N % 96 == 0 && N % 64 != 0) return {128, 96};
  const long t128 = (long)nimg * cdiv(HoWo, 128);
  const int pad128 = round_up(N, 128) - N;
  if (N >= 128 && pad128 * 20 <= N && t128 * cdiv(N, 128) >= 400) return {128, 128};
  if (t128 * cdiv(N, 64) >= 384) return {128, 64};
  return {64, 64};
}

// Runs the convolution; returns the tile choice (BM drives the statistics group layout).
template <int MODE, class Epi>
TileChoice conv_dispatch(const ConvShape& s, Epi ep, hipStream_t st);

}  // namespace atdn
