// Stem of the GMA encoders on the split-f16 engine: conv 7x7 stride 2, 3 -> 64 channels, + norm + ReLU
// (whl:GMA/core/extractor.py:122-123,161-163: conv1, norm1, relu1 of BasicEncoder).
//
// The layer is 1 % of the network's arithmetic and, as a ROW-mode layer on the exact-fp32 matrix core, took 320 us per
// 8 frames (v_mfma_f32_32x32x2_f32 is 16x slower per FLOP than the f16 shapes): as much as four of the 64-channel 3x3
// convolutions behind it. Here it is an implicit GEMM with K = 7 filter rows x (7 taps x 4 channels padded to 32):
//   * a block owns an 8 x 32 tile of OUTPUT pixels; its 21 x 70-pixel input patch (NHWC4 fp32, true zero padding) is
//     split into f16 hi / lo planes in LDS once. For filter row ky and K sub-step t a lane's eight operands are TWO
//     adjacent input pixels (taps kx = 4t + 2h, 4t + 2h + 1, four channels each) = one 16-byte run of a plane, and
//     the 32 output pixels of an MFMA row tile are 16 bytes apart: one conflict-free ds_read_b128 per operand.
//     (The 8th tap and the 4th channel hit zero weights.)
//   * weights: fragment-major split-f16 copy (pack_stem_sf, weights.h), 1 KiB per wave load, L2-resident (56 KB).
//   * MODE_RELU  (context network: BatchNorm folded into weights and bias): relu(conv + bias) -> sf.
//     InstanceNorm (feature network) needs the statistics of the whole image before the first output can be written.
//     The conv is so cheap that it is simply run TWICE: MODE_STATS computes it for the per-(32-pixel group, channel)
//     partial sums only (nothing else is written), MODE_NORM recomputes it and writes relu((conv + bias - mean) * rstd)
//     as sf. The raw fp32 tensor (237 MB per 8 frames), its re-read and the separate normalisation pass are gone.
//     Round 5, MODE_RAW_STATS (3): ONE pass that stores the raw conv + bias as fp32 (the same 4 bytes per value as the sf
//     tensor MODE_NORM wrote) and the same partial sums; nobody materialises relu(IN(.)) any more — layer1.0's first
//     conv normalises in its patch loader (conv_sf6.h, as every second conv of a block already does) and the block's
//     residual pass normalises the shortcut on the fly (in_apply_sf_kernel). The split-f16 feature network runs this;
//     the f16 fast mode keeps the two-pass form.
//   * the sf output of a row tile (32 pixels x 256 B, contiguous in memory) is assembled in a wave-private LDS slab and
//     stored as whole lines, 1 KiB per wave instruction.
#include "conv_mfma.h"
#include "kernels.h"
#include "sf.h"

namespace atdn {
namespace {

constexpr int TH = 8, TW = 32;
constexpr int PR = 2 * TH + 5;           // patch rows
constexpr int PC = 2 * TW + 6;           // patch pixels per row (one more than the taps reach: the padded 8th tap)
constexpr int PITCH = PC * 8;            // bytes per patch row of one plane (4 channels x f16)
constexpr int PLANE = PR * PITCH;
constexpr int SLAB_PITCH = 272;          // 256 B of one pixel's 64 sf channels + 16
constexpr int SLAB = 32 * SLAB_PITCH;
constexpr int LDS_BYTES = (2 * PLANE > 4 * SLAB) ? 2 * PLANE : 4 * SLAB;

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

struct StemArgs {
  const float4* img; int nimg, H, W, Ho, Wo, tiles_x, tiles_img;
  const float* wfrag; float wscale; const float* bias;
  float* out;                            // sf [nimg][Ho*Wo][64]
  float* part_sum; float* part_m2; float* part_cnt;   // MODE_STATS
  const float* mean; const float* rstd;  // MODE_NORM: [nimg][64]
};

template <int MODE>
__global__ __launch_bounds__(256, MODE == 1 ? 4 : 3) void stem_sf_kernel(const StemArgs a) {
  // (the statistics pass has no output slabs: four blocks per CU hide each other's patch loads)
  __shared__ __attribute__((aligned(16))) char lds[MODE == 1 ? 2 * PLANE : LDS_BYTES];
  __shared__ __attribute__((aligned(16))) float cst[3][64];   // bias, mean, rstd of this image (read per channel run in the epilogue)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int nblk = a.nimg * a.tiles_img;
  const int id = xcd_remap(blockIdx.x, nblk);
  const int img = id / a.tiles_img, tloc = id - img * a.tiles_img;
  const int ty0 = (tloc / a.tiles_x) * TH, tx0 = (tloc % a.tiles_x) * TW;

  if constexpr (MODE != 1) {
    if (tid < 64) cst[0][tid] = a.bias[tid];
    if constexpr (MODE == 2) {
      if (tid >= 64 && tid < 128) cst[1][tid - 64] = a.mean[img * 64 + tid - 64];
      if (tid >= 128 && tid < 192) cst[2][tid - 128] = a.rstd[img * 64 + tid - 128];
    }
  }
  // ---- input patch -> hi / lo planes
  {
    const float4* src = a.img + (long)img * a.H * a.W;
    const int iy0 = 2 * ty0 - 3, ix0 = 2 * tx0 - 3;
    constexpr int NLD = (PR * PC + 255) / 256;
    float4 v[NLD];
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int p = tid + 256 * k;
      const int py = p / PC, px = p - py * PC;
      const int iy = iy0 + py, ix = ix0 + px;
      const bool ok = p < PR * PC && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      v[k] = keep_if(ok, src[ok ? (long)iy * a.W + ix : 0]);
    }
    bool clamped = false;
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int p = tid + 256 * k;
      if (p < PR * PC) {   // (round 5: the packed split of sf.h — two pair conversions, four v_fma_mix residuals, one range test)
        float4 x = v[k];
        const float m = fmaxf(fmaxf(fabsf(x.x), fabsf(x.y)), fmaxf(fabsf(x.z), fabsf(x.w)));
        clamped |= !(m <= 65504.f) | __builtin_isunordered(x.x, x.y) | __builtin_isunordered(x.z, x.w);
        x.x = __builtin_amdgcn_fmed3f(x.x, -65504.f, 65504.f); x.y = __builtin_amdgcn_fmed3f(x.y, -65504.f, 65504.f);
        x.z = __builtin_amdgcn_fmed3f(x.z, -65504.f, 65504.f); x.w = __builtin_amdgcn_fmed3f(x.w, -65504.f, 65504.f);
        const unsigned h0 = sf_cvt_pk_(x.x, x.y), h1 = sf_cvt_pk_(x.z, x.w);
        const unsigned l0 = sf_residual_pk_(h0, x.x, x.y), l1 = sf_residual_pk_(h1, x.z, x.w);
        *reinterpret_cast<uint2*>(lds + p * 8) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(lds + PLANE + p * 8) = make_uint2(l0, l1);
      }
    }
    sf_report(clamped);
  }
  __syncthreads();

  // ---- K loop: 7 filter rows x 2 sub-steps; wave = output rows 2*wave, 2*wave + 1 x all 64 channels
  constexpr bool SWAP = MODE != 1;       // weights as the MFMA row operand: a lane ends up with 16 channels of ONE pixel
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const char* abase = lds + (4 * wave) * PITCH + (2 * r + 2 * h) * 8;
  const char* wbase = reinterpret_cast<const char*>(a.wfrag) + lane * 16;
#pragma unroll
  for (int q = 0; q < 14; ++q) {
    const int ky = q >> 1, t = q & 1;
    f16x8 wh[2], wl[2], ah[2], al[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      wh[j] = *reinterpret_cast<const f16x8*>(wbase + ((j * 14 + q) * 2) * 1024);
      wl[j] = *reinterpret_cast<const f16x8*>(wbase + ((j * 14 + q) * 2 + 1) * 1024);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      ah[i] = *reinterpret_cast<const f16x8*>(abase + (2 * i + ky) * PITCH + 32 * t);
      al[i] = *reinterpret_cast<const f16x8*>(abase + PLANE + (2 * i + ky) * PITCH + 32 * t);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if constexpr (SWAP) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[j], al[i], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[j], ah[i], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[j], ah[i], acc[i][j], 0, 0, 0);
        } else {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], wh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], wl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], wh[j], acc[i][j], 0, 0, 0);
        }
      }
  }

  if constexpr (MODE == 1) {
    // ---- statistics: lane (r, h) holds channel 32 j + r of the 16 pixels (e & 3) + 8 (e >> 2) + 4 h of row tile i
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int oy = ty0 + 2 * wave + i;
      const int grp = tloc * TH + 2 * wave + i;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = 32 * j + r;
        const float bias = a.bias[n];
        float v[16];
        float sum = 0.f;
        int cnt = 0;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int ox = tx0 + (e & 3) + 8 * (e >> 2) + 4 * h;
          v[e] = acc[i][j][e] * a.wscale + bias;
          if (oy < a.Ho && ox < a.Wo) { sum += v[e]; ++cnt; }
        }
        sum += __shfl_xor(sum, 32);
        cnt += __shfl_xor(cnt, 32);
        const float mean = sum / (float)(cnt > 0 ? cnt : 1);
        float m2 = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int ox = tx0 + (e & 3) + 8 * (e >> 2) + 4 * h;
          if (oy < a.Ho && ox < a.Wo) { const float d = v[e] - mean; m2 += d * d; }
        }
        m2 += __shfl_xor(m2, 32);
        const long gi = (long)img * (a.tiles_img * TH) + grp;
        if (h == 0) {
          a.part_sum[gi * 64 + n] = sum;
          a.part_m2[gi * 64 + n] = m2;
        }
        if (lane == 0 && j == 0) a.part_cnt[gi] = (float)cnt;
      }
    }
  } else {
    // ---- lane (r, h) holds pixel tx0 + r of row tile i, channels 32 j + 8 k + 4 h + (0..3) in registers 4k..4k+3
    __syncthreads();                     // every wave is done with the patch planes: the slabs reuse them
    char* slab = lds + wave * SLAB;
    bool clamped = false;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int oy = ty0 + 2 * wave + i;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int c = 32 * j + 8 * k + 4 * h;
          const float4 b = *reinterpret_cast<const float4*>(&cst[0][c]);
          float x[4] = {acc[i][j][4 * k] * a.wscale + b.x, acc[i][j][4 * k + 1] * a.wscale + b.y,
                        acc[i][j][4 * k + 2] * a.wscale + b.z, acc[i][j][4 * k + 3] * a.wscale + b.w};
          if constexpr (MODE == 3) {   // raw fp32 into the slab: 64 channels x 4 B = the 256 B of an sf pixel
            *reinterpret_cast<float4*>(slab + r * SLAB_PITCH + c * 4) = make_float4(x[0], x[1], x[2], x[3]);
            continue;
          }
          if constexpr (MODE == 2) {
            const float4 mu = *reinterpret_cast<const float4*>(&cst[1][c]);
            const float4 rs = *reinterpret_cast<const float4*>(&cst[2][c]);
            x[0] = (x[0] - mu.x) * rs.x; x[1] = (x[1] - mu.y) * rs.y; x[2] = (x[2] - mu.z) * rs.z; x[3] = (x[3] - mu.w) * rs.w;
          }
          // ReLU and the format's clamp in one v_med3 per value; the range test on the largest of the four (a NaN: the
          // unordered compares — v_max and v_med3 drop it)
          const float m = fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3]));
          clamped |= !(m <= 65504.f) | __builtin_isunordered(x[0], x[1]) | __builtin_isunordered(x[2], x[3]);
#pragma unroll
          for (int e = 0; e < 4; ++e) x[e] = __builtin_amdgcn_fmed3f(x[e], 0.f, 65504.f);
          const unsigned h0 = sf_cvt_pk_(x[0], x[1]), h1 = sf_cvt_pk_(x[2], x[3]);
          const unsigned l0 = sf_residual_pk_(h0, x[0], x[1]), l1 = sf_residual_pk_(h1, x[2], x[3]);
          char* d = slab + r * SLAB_PITCH + j * 128 + (8 * k + 4 * h) * 2;
          *reinterpret_cast<uint2*>(d) = make_uint2(h0, h1);
          *reinterpret_cast<uint2*>(d + 64) = make_uint2(l0, l1);
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      // 16 lanes per pixel: one 1 KiB store covers 4 pixels (contiguous in memory)
      float* orow = a.out + ((long)img * a.Ho * a.Wo + (long)oy * a.Wo + tx0) * 64;
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const int px = 4 * p + (lane >> 4);
        const v4f v = *reinterpret_cast<const v4f*>(slab + px * SLAB_PITCH + (lane & 15) * 16);
        if (oy < a.Ho && tx0 + px < a.Wo) *reinterpret_cast<v4f*>(orow + px * 64 + (lane & 15) * 4) = v;
      }
      if constexpr (MODE == 3) {
        // statistics of the row tile's 32 pixels from the slab: lane = channel, the pixels in order (the grouping and the order
        // follow from the layer's geometry alone: a clip, a continued clip and single pairs produce the same bits)
        float v[32];
        float sum = 0.f;
        int cnt = 0;
#pragma unroll
        for (int px = 0; px < 32; ++px) {
          v[px] = *reinterpret_cast<const float*>(slab + px * SLAB_PITCH + lane * 4);
          if (oy < a.Ho && tx0 + px < a.Wo) { sum += v[px]; ++cnt; }
        }
        const float mean = sum / (float)(cnt > 0 ? cnt : 1);
        float m2 = 0.f;
#pragma unroll
        for (int px = 0; px < 32; ++px)
          if (oy < a.Ho && tx0 + px < a.Wo) { const float d = v[px] - mean; m2 += d * d; }
        const long gi = (long)img * (a.tiles_img * TH) + tloc * TH + 2 * wave + i;
        a.part_sum[gi * 64 + lane] = sum;
        a.part_m2[gi * 64 + lane] = m2;
        if (lane == 0) a.part_cnt[gi] = (float)cnt;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();   // the slab is rewritten by the next row tile
    }
    sf_report(clamped);
  }
}

}  // namespace

int stem_sf_groups(int Ho, int Wo) { return cdiv(Ho, TH) * cdiv(Wo, TW) * TH; }

void launch_stem_sf(int mode, const float* img4, int nimg, int H, int W, const float* wfrag, float wscale,
                    const float* bias, float* out_sf, float* part_sum, float* part_m2, float* part_cnt,
                    const float* mean, const float* rstd, hipStream_t st) {
  StemArgs a{};
  a.img = reinterpret_cast<const float4*>(img4); a.nimg = nimg; a.H = H; a.W = W;
  a.Ho = conv_out(H, 7, 2, 3); a.Wo = conv_out(W, 7, 2, 3);
  a.tiles_x = cdiv(a.Wo, TW); a.tiles_img = a.tiles_x * cdiv(a.Ho, TH);
  a.wfrag = wfrag; a.wscale = wscale; a.bias = bias; a.out = out_sf;
  a.part_sum = part_sum; a.part_m2 = part_m2; a.part_cnt = part_cnt; a.mean = mean; a.rstd = rstd;
  ATDN_CHECK(wfrag != nullptr && bias != nullptr, "stem weights missing");
  const dim3 grid(nimg * a.tiles_img), block(256);
  if (mode == 0) {
    ATDN_CHECK(out_sf != nullptr, "stem output missing");
    hipLaunchKernelGGL(stem_sf_kernel<0>, grid, block, 0, st, a);
  } else if (mode == 1) {
    ATDN_CHECK(part_sum && part_m2 && part_cnt, "stem statistics buffers missing");
    hipLaunchKernelGGL(stem_sf_kernel<1>, grid, block, 0, st, a);
  } else if (mode == 3) {
    ATDN_CHECK(out_sf && part_sum && part_m2 && part_cnt, "stem raw output / statistics buffers missing");
    hipLaunchKernelGGL(stem_sf_kernel<3>, grid, block, 0, st, a);
  } else {
    ATDN_CHECK(out_sf && mean && rstd, "stem normalisation operands missing");
    hipLaunchKernelGGL(stem_sf_kernel<2>, grid, block, 0, st, a);
  }
  ATDN_HIP(hipGetLastError());
}

}  // namespace atdn
