// Persistent LSTM scan of the CLVO head (atdn_vslam/odometry/network.py:137-140: lstm1 -> lstm_linear (+ Mish) -> lstm2, the state
// never reset inside a sequence: evaluate_odometry.py:60-75) — SURVEY K17: ONE launch per sequence instead of one launch per time
// step. Batch row count 1 (a sequence is scanned in order; the batched per-frame forward keeps the per-step kernel of kernels.hip).
//
// What the per-step form costs: 4.4-5.0 us per step = a dependent kernel boundary (~1.5 us) + 13.6 MB of recurrent weights streamed
// from L2 by 4.6 k waves, every step again: 20-21 ms per KITTI-00 sequence, replicated on every rank (the only part of a sharded
// sequence that does not shrink with the number of GPUs). Here the weights are read ONCE: 128 workgroups of 4 arithmetic waves, one
// hidden unit per wave, each wave keeps the 13 weight rows of its unit in registers for the whole sequence —
//     W_hh1[g*512 + u][:] (4 gates), W_lin[u][:], W_ih2[g*512 + u][:] (4), W_hh2[g*512 + u][:] (4): 13 x 512 floats = 104 per lane
// — and a time step is one TICK of the same three-stage software pipeline the per-step kernel runs (tick s: lstm1 for step s,
// lstm_linear for step s - 1, lstm2 for step s - 2; every stage reads what tick s - 1 produced):
//     sweep    waves 0-2 poll the 3 x 512 results (h1 | lin | h2) the workgroups published in tick s - 1 and put them in LDS
//     prefetch a fifth wave keeps lstm1's input-projection terms six ticks ahead, global -> LDS directly (LDS-DMA)
//     barrier
//     dot      every arithmetic wave: 13 rows x 512 from registers (packed FMAs) x the vectors from LDS, 9 wave reductions with
//              their DPP chains interleaved, ALL gate functions of the tick in one v_exp_f32 + v_rcp_f32 pair (one value per lane)
//     barrier, publish: the workgroup's 12 results as ONE store instruction (tag = tick + 1)
// The exchange is the data-tagged granule hand-off of cdna_hip_programming.md section 6 Guideline 16 (R2): every value travels as
// ONE naturally aligned 8-byte {tag, value} written by a relaxed agent-scope store (sc1, write-through) and read by relaxed
// agent-scope loads (sc1: never served from this CU's L1) until every tag matches — the data is the flag, no fence on either side.
// Ring of two slots by tick parity: a workgroup can overwrite slot s & 1 (tick s + 2) only after it has swept tick s + 1, which
// every workgroup publishes only after ITS sweep of tick s has completed. All polled words are zeroed by a memset node in front
// of every launch (tag 0 never matches: tags start at 1).
// Where a tick's 2.0 us go (tools/scan_time.py, ablation modes; profiles/r06_scan_*.txt): barriers + publish 0.4, arithmetic
// 0.55, the exchange 1.1 (store -> visible ~0.45 + one polling pass ~0.6). Steps on the way (us per tick): first version, every
// wave storing its own granules and loading its own input-projection terms 3.09 | one store per workgroup, terms by LDS-DMA 2.85
// | gate batch, packed FMAs, interleaved reductions 2.55-2.68 | first poll delayed until the data is about to be visible 2.02.
// Residency: 128 workgroups of 320 threads on 256 CUs — resident together whenever nothing else holds the chip for good; every
// spin is bounded (a workgroup that gives up sets the abort word, every other one sees it in its next pass, all of them poison the
// output with NaN and exit: a failed scan is loud, and the host falls back to the per-step kernel for that handle afterwards).
#include "kernels.h"
#include <cstdlib>

namespace atdn {
namespace {

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned int gu32;

constexpr int HD = 512;            // hidden size of both cells
constexpr unsigned long long SPIN_LIMIT_TICKS = 50000000ull;   // 0.5 s of the 100 MHz real-time clock (s_memrealtime), then give up

// sum over the 64 lanes on the vector ALU (DPP row shifts + row broadcasts; no LDS crossbar), returned as a wave-uniform value
__device__ __forceinline__ float wave_sum_dpp(float v) {
#define ATDN_DPP_ADD(ctrl, rmask) \
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, rmask, 0xF, false))
  ATDN_DPP_ADD(0x111, 0xF);   // row_shr:1
  ATDN_DPP_ADD(0x112, 0xF);   // row_shr:2
  ATDN_DPP_ADD(0x114, 0xF);   // row_shr:4
  ATDN_DPP_ADD(0x118, 0xF);   // row_shr:8   -> lane 15 of every row holds the row's sum
  ATDN_DPP_ADD(0x142, 0xA);   // row_bcast:15 into rows 1 and 3
  ATDN_DPP_ADD(0x143, 0xC);   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
#undef ATDN_DPP_ADD
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// 8 (or 16) products of a row's slice as packed FMAs (v_pk_fma_f32: two per instruction)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk2(float a, float b) { f32x2 v; v.x = a; v.y = b; return v; }
__device__ __forceinline__ f32x2 dot8acc(f32x2 acc, const float4& wa, const float4& wb, const float4& xa, const float4& xb) {
  acc = __builtin_elementwise_fma(pk2(wa.x, wa.y), pk2(xa.x, xa.y), acc);
  acc = __builtin_elementwise_fma(pk2(wa.z, wa.w), pk2(xa.z, xa.w), acc);
  acc = __builtin_elementwise_fma(pk2(wb.x, wb.y), pk2(xb.x, xb.y), acc);
  acc = __builtin_elementwise_fma(pk2(wb.z, wb.w), pk2(xb.z, xb.w), acc);
  return acc;
}
__device__ __forceinline__ float dot8(const float4& wa, const float4& wb, const float4& xa, const float4& xb) {
  const f32x2 acc = dot8acc(pk2(0.f, 0.f), wa, wb, xa, xb);
  return acc.x + acc.y;
}
__device__ __forceinline__ float dot8b(const float4& wa, const float4& wb, const float4& xa, const float4& xb, const float4& va,
                                       const float4& vb, const float4& ya, const float4& yb) {
  const f32x2 acc = dot8acc(dot8acc(pk2(0.f, 0.f), wa, wb, xa, xb), va, vb, ya, yb);
  return acc.x + acc.y;
}

// Mish(x) = x tanh(log(1 + e^x)) = x t / (t + 2), t = e^x (e^x + 2): one v_exp_f32 and one v_rcp_f32 (as the head's CNN tails)
__device__ __forceinline__ float mish_fast_(float x) {
  const float e = __builtin_amdgcn_exp2f(fminf(x, 20.0f) * 1.4426950408889634f);
  const float t = e * (e + 2.0f);
  return x > 20.0f ? x : x * t * __builtin_amdgcn_rcpf(t + 2.0f);
}

struct ScanArgs {
  const float* pre1;                 // [T][4 * 512]: W_ih1 x_t + b_ih1 of every step (one GEMM in front of the scan)
  const float *Whh1, *bhh1, *Wlin, *blin, *Wih2, *bih2, *Whh2, *bhh2;
  float* state;                      // [4][512]: h1, c1, h2, c2 — read at the start, written at the end
  float* h2seq;                      // [T][512]: lstm2's hidden state after every step (input of the regressors)
  unsigned long long* xch;           // [2 slots][workgroup][h1 | lin | h2][unit of the workgroup]: 3 x 512 granules per slot
  unsigned int* abort_word;          // behind the granules
  int T;
  int mode;                          // diagnostics (ATDN_SCAN_MODE): bits 0-3 initial poll delay, 16 no arithmetic, 32 no sweep (both: timing only), 64 the 64-workgroup form
};

constexpr int PD = 6;                // ticks the input-projection terms are fetched ahead (LDS-DMA ring of 8 slots)
constexpr int NGRAN = 3 * HD;        // granules per slot: [workgroup][h1 | lin | h2][unit of the workgroup]

__device__ __forceinline__ void barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// NWV = hidden units (= arithmetic waves) per workgroup; 512 / NWV workgroups of NWV + 1 waves (the last one prefetches)
template <int NWV>
__global__ __launch_bounds__((NWV + 1) * 64) void lstm_scan_kernel(const ScanArgs a) {
  constexpr int GPW = 3 * NWV;
  static_assert(NWV == 4 || NWV == 8, "operand runs of four units must stay inside one workgroup's block");
  // ONE LDS object. Image of a tick's inputs in the order they were published: [workgroup][h1 | lin | h2][unit of the workgroup]
  __shared__ __attribute__((aligned(16))) float lds[2 * NGRAN + 8 * 32 + 32 + 4];
  float* p1s = lds + 2 * NGRAN;                                             // [8 ring slots][gate 4][unit NWV] (32 floats apart)
  float* pub = p1s + 8 * 32;                                                // [h1 | lin | h2][unit NWV] of this tick
  int* dead = reinterpret_cast<int*>(pub + 32);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int T = a.T;

  if (wave == NWV) {
    // ---- the prefetch wave: lstm1's input-projection terms of tick s + PD, global -> LDS directly (LDS-DMA: no register, no
    // wait until the data is used). A wave of its own because the compiler drains vmcnt(0) in front of every LDS read of a wave
    // that has an LDS-DMA in flight (cdna_hip_programming.md section 5, 'Pipelining across barriers'): this path reads LDS only
    // through inline assembly (the abort flag), waits with a COUNTED vmcnt, and joins the other waves' two barriers per tick.
    const float* src = a.pre1 + (lane / NWV) * HD + blockIdx.x * NWV + (lane % NWV);   // lane = gate * NWV + unit of the workgroup
    auto fetch_p1 = [&](int t) __attribute__((always_inline)) {
      const int tt = min(t, T - 1);
      if (lane < 4 * NWV)
        __builtin_amdgcn_global_load_lds(src + (long)tt * (4 * HD), (__attribute__((address_space(3))) void*)(p1s + (t & 7) * 32), 4, 0, 0);
    };
#pragma unroll
    for (int t = 0; t < PD; ++t) fetch_p1(t);
    barrier_lds();
    if (a.mode & 128) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return; }   // test hook: the launch gives up at once
    const unsigned dead_addr = (unsigned)(unsigned long)(__attribute__((address_space(3))) int*)dead;
    for (int s = 0; s < T + 2; ++s) {
      fetch_p1(s + PD);
      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PD) : "memory");   // tick s's terms (requested PD ticks ago) have landed
      barrier_lds();
      int d;
      asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(d) : "v"(dead_addr) : "memory");
      if (__builtin_amdgcn_readfirstlane(d)) break;
      barrier_lds();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA write may outlive the workgroup
    return;
  }

  const int u = blockIdx.x * NWV + wave;
  gu64* xch = (gu64*)a.xch;
  gu32* abort_word = (gu32*)a.abort_word;
  // ---- this unit's weight rows -> registers (lane l: columns 4 l .. 4 l + 3 and 256 + 4 l .. 256 + 4 l + 3 of every row)
  float4 w1[4][2], wl[2], wi[4][2], wh[4][2];
  auto row2 = [&](const float* W, int row, float4* dst) {
    const float4* p = reinterpret_cast<const float4*>(W + (long)row * HD);
    dst[0] = p[lane]; dst[1] = p[64 + lane];
  };
#pragma unroll
  for (int g = 0; g < 4; ++g) { row2(a.Whh1, g * HD + u, w1[g]); row2(a.Wih2, g * HD + u, wi[g]); row2(a.Whh2, g * HD + u, wh[g]); }
  row2(a.Wlin, u, wl);
  // per-lane constants of the gate batch: lane 0-3 lstm1's gates, 4 lstm_linear, 5-8 lstm2's gates
  float bvec = 0.f;
  if (lane < 4) bvec = a.bhh1[lane * HD + u];
  else if (lane == 4) bvec = a.blin[u];
  else if (lane < 9) bvec = a.bih2[(lane - 5) * HD + u] + a.bhh2[(lane - 5) * HD + u];
  const bool istanh = lane == 2 || lane == 7;
  const float kvec = lane > 8 ? 0.f : lane == 4 ? 1.4426950408889634f : istanh ? -2.8853900817779268f : -1.4426950408889634f;
  const int p1idx = min(lane, 3) * NWV + wave;     // lane g < 4: lstm1's input-projection term of gate g
  float c1 = a.state[1 * HD + u], c2 = a.state[3 * HD + u];
  float h1_last = a.state[0 * HD + u], h2_last = a.state[2 * HD + u];
  // the state in front of the sequence plays tick -1's results: h1 for tick 0 (parity 0), h2 for tick 2 (parity 0)
  for (int k = threadIdx.x; k < HD; k += NWV * 64) {
    lds[(k / NWV) * GPW + (k % NWV)] = a.state[0 * HD + k];
    lds[(k / NWV) * GPW + 2 * NWV + (k % NWV)] = a.state[2 * HD + k];
  }
  if (threadIdx.x == 0) *dead = 0;
  // sweepers (waves 0-2): lane's granule i of a slot is flat index (8 wave + i) 64 + lane, the same index in the LDS image;
  // bit i of h2mask: that granule is an h2 value (not taken before tick 3: the sequence's initial h2 serves tick 2)
  unsigned h2mask = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) h2mask |= ((((wave % 3) * 8 + i) * 64 + lane) % GPW >= 2 * NWV ? 1u : 0u) << i;
  // operand addresses in the image: units 4 l .. 4 l + 3 (workgroup 4 l / NWV, offset 4 l % NWV) and 256 + the same
  const int xo = ((4 * lane) / NWV) * GPW + (4 * lane) % NWV;
  constexpr int XB = (256 / NWV) * GPW;   // the same units + 256
  barrier_lds();

  bool failed = (a.mode & 128) != 0;   // ATDN_SCAN_TEST_ABORT: behave like a launch whose spins ran out
  if (failed && threadIdx.x == 0) __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int dly = (a.mode & 15) ? (a.mode & 15) : 8, streak = 0;   // sweepers: delay of the first poll in units of 128 clocks (adaptive)
  for (int s = 0; s < T + 2 && !failed; ++s) {
    const int par = s & 1;
    float* img = lds + par * NGRAN;
    const bool doA = s < T, doC = s >= 2;   // (lstm_linear is live in ticks 1 .. T; nothing of it is carried)
    // ---- sweep: everything tick s - 1 published carries tag s
    if (wave < 3 && s >= 1 && !(a.mode & 32)) {
      gu64* g = xch + (long)((s - 1) & 1) * NGRAN + wave * 512 + lane;
      unsigned v[8];
      // The first pass is held back until the stores of tick s - 1 are about to be visible: a pass that comes too early costs a
      // whole round trip AND stands in the fabric's queues in front of the pass that would have succeeded (128 workgroups x 12 KB
      // per pass). Measured, 4,540 ticks, 128 workgroups: no delay 2.62 us per tick, 512 clocks 2.35, 768 clocks 2.25, 1,024
      // clocks 2.02, then +0.05 us per further 128 clocks (profiles/r06_scan_poll_delay.txt). The delay follows the box: one unit
      // (128 clocks) more after a tick whose first pass failed, one less after 32 ticks in a row whose first pass succeeded.
      for (int z = 0; z < dly; ++z) __builtin_amdgcn_s_sleep(2);
      unsigned long long t_start = 0;
      for (unsigned spins = 0;; ++spins) {
        bool ok = true;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const unsigned long long x = __hip_atomic_load(g + 64 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          v[i] = (unsigned)x;
          ok &= (unsigned)(x >> 32) == (unsigned)s;
        }
        if (__all(ok)) {
          if (spins == 0) { if (++streak == 32) { streak = 0; dly = max(dly - 1, 0); } }
          else { streak = 0; dly = min(dly + 1, 24); }
          break;
        }
        const unsigned ab = __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // bounded by WALL time, not by a pass count: a workgroup that waits for a CU behind another stream's kernels is late, not
        // lost (the clock is read on the first failed pass and then every 64th)
        bool late = false;
        if (spins == 0) t_start = __builtin_amdgcn_s_memrealtime();
        else if ((spins & 63u) == 0u) late = __builtin_amdgcn_s_memrealtime() - t_start > SPIN_LIMIT_TICKS;
        if (ab != 0u || late) {   // uniform: every lane read the same word, the clock is a scalar
          if (lane == 0) { __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); *dead = 1; }
          break;
        }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (!((h2mask >> i) & 1u) || s >= 3) img[wave * 512 + 64 * i + lane] = __uint_as_float(v[i]);
    }
    barrier_lds();
    if (*dead) { failed = true; break; }
    // ---- the three stages of this tick for unit u: 13 dot products (packed FMAs), 9 sums reduced over the wave with their DPP
    // chains interleaved, then ALL gate functions of the tick evaluated at once, one value per lane:
    //   lane 0-3: i, f, g, o of lstm1 | lane 4: lstm_linear's Mish | lane 5-8: i, f, g, o of lstm2
    // sigmoid(x) = 1 / (1 + e^-x), tanh(x) = 2 sigmoid(2x) - 1, Mish(x) = x t / (t + 2) with t = e^x (e^x + 2): ONE v_exp_f32 and
    // ONE v_rcp_f32 for the nine of them, one more pair for the two tanh(c'). (Every lane evaluating every function itself, as
    // the first version of this kernel did, spent 24 quarter-rate instructions per tick and wave.)
    // Stages that are not live in this tick (pipeline fill and drain) compute on whatever the image holds and are discarded.
    float h1n = 0.f, linv = 0.f, h2n = 0.f;
    if (!(a.mode & 16)) {
      const float4 xa = *reinterpret_cast<const float4*>(img + xo), xb = *reinterpret_cast<const float4*>(img + XB + xo);
      const float4 ya = *reinterpret_cast<const float4*>(img + NWV + xo), yb = *reinterpret_cast<const float4*>(img + XB + NWV + xo);
      const float4 za = *reinterpret_cast<const float4*>(img + 2 * NWV + xo), zb = *reinterpret_cast<const float4*>(img + XB + 2 * NWV + xo);
      const float p1v = p1s[(s & 7) * 32 + p1idx];
      float sv[9];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        sv[g] = dot8(w1[g][0], w1[g][1], xa, xb);
        sv[5 + g] = dot8b(wi[g][0], wi[g][1], ya, yb, wh[g][0], wh[g][1], za, zb);
      }
      sv[4] = dot8(wl[0], wl[1], xa, xb);
#define ATDN_DPP_STEP(ctrl, rmask) _Pragma("unroll") for (int k = 0; k < 9; ++k) \
        sv[k] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sv[k]), ctrl, rmask, 0xF, false))
      ATDN_DPP_STEP(0x111, 0xF); ATDN_DPP_STEP(0x112, 0xF); ATDN_DPP_STEP(0x114, 0xF); ATDN_DPP_STEP(0x118, 0xF);
      ATDN_DPP_STEP(0x142, 0xA); ATDN_DPP_STEP(0x143, 0xC);
#undef ATDN_DPP_STEP
      // the totals (lane 63) as scalars FIRST, pinned in uniform control flow: written as `lane == k ? readlane(..) : gv` the
      // compiler turned the selects into branches and sank a total's last DPP addition into its branch, where only lane k
      // executes it — lane 63, the one the readlane reads, kept the sum without it
      int tot[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        tot[k] = __builtin_amdgcn_readlane(__float_as_int(sv[k]), 63);
        asm volatile("" : "+s"(tot[k]));
      }
      float gv = 0.f;
#pragma unroll
      for (int k = 0; k < 9; ++k) gv = lane == k ? __int_as_float(tot[k]) : gv;
      const float x = (gv + bvec) + (lane < 4 ? p1v : 0.f);
      const float e = __builtin_amdgcn_exp2f((lane == 4 ? fminf(x, 20.0f) : x) * kvec);
      const float t = e * (e + 2.0f);
      const float r = __builtin_amdgcn_rcpf(lane == 4 ? t + 2.0f : 1.0f + e);
      const float out = lane == 4 ? (x > 20.0f ? x : x * t * r) : istanh ? __builtin_fmaf(2.0f, r, -1.0f) : r;
      auto at = [&](float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); };
      const float cnA = at(out, 1) * c1 + at(out, 0) * at(out, 2);
      const float cnC = at(out, 6) * c2 + at(out, 5) * at(out, 7);
      const float y = lane == 0 ? cnA : cnC;
      const float th = __builtin_fmaf(2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(y * -2.8853900817779268f)), -1.0f);
      h1n = at(out, 3) * at(th, 0);
      linv = at(out, 4);
      h2n = at(out, 8) * at(th, 1);
      if (doA) { c1 = cnA; h1_last = h1n; }
      if (doC) { c2 = cnC; h2_last = h2n; }
    }
    // ---- publish: the workgroup's 24 results as ONE store instruction of wave 4 (tag = tick + 1), and the output row
    if (lane == 0) { pub[wave] = h1n; pub[NWV + wave] = linv; pub[2 * NWV + wave] = h2n; }
    barrier_lds();
    if (wave == NWV - 1 && lane < GPW && s <= T) {
      const unsigned long long tag = (unsigned long long)(unsigned)(s + 1) << 32;
      __hip_atomic_store(xch + (long)par * NGRAN + blockIdx.x * GPW + lane, tag | __float_as_uint(pub[lane]), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    }
    if (wave == NWV - 1 && lane < NWV && doC) a.h2seq[(long)(s - 2) * HD + blockIdx.x * NWV + lane] = pub[2 * NWV + lane];
  }
  if (failed) {   // loud failure: every row of this workgroup's units is NaN (the regressors turn that into NaN poses)
    const float nan = __uint_as_float(0x7FC00000u);
    for (int t = lane; t < T; t += 64) a.h2seq[(long)t * HD + u] = nan;
    h1_last = c1 = h2_last = c2 = nan;
  }
  if (lane == 0) {
    a.state[0 * HD + u] = h1_last; a.state[1 * HD + u] = c1; a.state[2 * HD + u] = h2_last; a.state[3 * HD + u] = c2;
  }
}

}  // namespace

// Can every workgroup of the persistent kernel be resident at once on this device? (A partitioned or smaller part — fewer CUs than
// the 128 workgroups need at one 5-wave workgroup per CU — would park the surplus workgroups behind the resident ones, which
// wait for them: the bounded spins would end the launch after 0.5 s with NaN poses. Such a device keeps the per-step kernel.)
bool lstm_scan_fits_device() {
  int dev = 0, cus = 0, per_cu = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, lstm_scan_kernel<4>, 5 * 64, 0) != hipSuccess) return false;
  // the occupancy query can over-report by one workgroup per CU (MI355X_MICROARCH.md, residency): count one fewer where it says > 1
  const int safe = per_cu > 1 ? per_cu - 1 : per_cu;
  return (long)safe * cus >= HD / 4;
}

long lstm_scan_exchange_bytes() { return (long)(2 * NGRAN) * 8 + 64; }

void launch_lstm_scan(const float* pre1, const float* Whh1, const float* bhh1, const float* Wlin, const float* blin,
                      const float* Wih2, const float* bih2, const float* Whh2, const float* bhh2, float* state, float* h2seq,
                      void* exchange, int T, hipStream_t st) {
  ATDN_CHECK(T >= 1 && exchange != nullptr, "lstm_scan: bad arguments");
  // every polled word (granule tags, abort word) is zeroed in front of EVERY launch: tags of an earlier sequence must not match
  ATDN_HIP(hipMemsetAsync(exchange, 0, (size_t)lstm_scan_exchange_bytes(), st));
  ScanArgs a{pre1, Whh1, bhh1, Wlin, blin, Wih2, bih2, Whh2, bhh2, state, h2seq,
             reinterpret_cast<unsigned long long*>(exchange),
             reinterpret_cast<unsigned int*>(reinterpret_cast<char*>(exchange) + (long)(2 * NGRAN) * 8), T,
             (getenv("ATDN_SCAN_MODE") ? atoi(getenv("ATDN_SCAN_MODE")) : 0) | (getenv("ATDN_SCAN_TEST_ABORT") ? 128 : 0)};
  // 128 workgroups of 4 units: one arithmetic wave per SIMD (64 of 8 units: two waves share a SIMD's vector ALU; 2.44 against
  // 2.02 us per tick at each form's best poll delay). ATDN_SCAN_MODE bit 6 selects the 64-workgroup form for comparison.
  if (a.mode & 64) hipLaunchKernelGGL(lstm_scan_kernel<8>, dim3(HD / 8), dim3(9 * 64), 0, st, a);
  else hipLaunchKernelGGL(lstm_scan_kernel<4>, dim3(HD / 4), dim3(5 * 64), 0, st, a);
  ATDN_HIP(hipGetLastError());
}

}  // namespace atdn
