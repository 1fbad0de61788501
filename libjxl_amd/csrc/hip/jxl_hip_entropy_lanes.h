// libjxl_amd — lane-parallel AC entropy decode for gfx950 (MI355X): the throughput form of the entropy stage.
//
// Replaces (like k_entropy_uni) lib/jxl/dec_group.cc:469-542,594-639 (DecodeACVarBlock / GetBlockFromBitstream),
// dec_ans.h:170-257 (rANS symbol + hybrid uint), ans_common.h:102-142 (alias lookup), ac_context.h:63-143.
//
// One AC section (256x256 group) is an inherently serial adaptive-context rANS stream, so the only parallelism is
// ACROSS sections. Here every LANE of a wave decodes its own section: the decoder state (rANS state, bit position,
// block / coefficient cursors) lives in VGPRs, the frame's entropy tables are shared in LDS by the workgroup, and one
// trip of the hot loop decodes one coefficient token for every runnable lane.
//
// What bounds it (round-3 measurements, scripts/r03_entropy_probe.py, scripts/ubench_trip.hip): a launch lasts as long as
// its busiest lane, i.e. (tokens of that lane's sections) x (cycles per trip) / (share of trips the lane takes part in).
// A lone wave issues one instruction per ~4 cycles, so the trip is written to be SHORT rather than parallel:
//   * one straight-line trip (~85 issue slots): the only branch is the wave-uniform skip of the extra-bits code;
//     everything rare (block transitions, the last partial coefficient chunk, the kend store, stream errors, ring
//     refills) happens in the service phase, which sees a finished lane as mode kFlush;
//   * per-frame parameters sit in scalar registers (read through the constant address space), so the loop contains no
//     flat / global load at all and no s_waitcnt vmcnt: gfx950 counts loads and stores in ONE counter, so any wait for a
//     load result is also a wait for every coefficient store before it. The stores are issued from inline assembly (the
//     compiler's wait insertion never sees them) and nothing in the kernel reads those locations back;
//   * the compressed stream and the block records reach their LDS rings by LDS-DMA (global_load_lds_dword: one ring row
//     for all lanes per instruction, each lane from its own stream position), issued in one service phase and counted
//     as landed in the next, so no service phase waits for HBM.
//
// Memory behaviour:
//   * each lane's compressed stream is staged through a 16-word ring in LDS ([slot][lane] layout: conflict-free; two
//     mirror rows so that a 3-word window never wraps); the hot loop extracts bits with ds_read2 + v_alignbit;
//   * coefficients are written in SCAN order (position k of the coefficient order, zeros included) as aligned 8-byte
//     chunks (four int16 / two int32 collected in a register pair: a lane's stores are uncoalesced by nature, and the
//     memory pipeline serialises such a wave-instruction per lane), plus the number of valid entries per
//     (block, channel) in `kend`; the transform kernels apply the order permutation when they stage a block into LDS.
//     No zero fill of the coefficient buffer, no order-table lookup and no dependent load in the decode loop.
//
// Control flow is a per-lane state machine kept convergent on the hot part:
//   RUN    a coefficient token is pending                        -> the hot trip (one shared code sequence)
//   FLUSH  the lane finished a (block, channel)                  -> service: last chunk + kend, then as WAIT
//   WAIT   block / channel transition pending (or the ring is low) -> serviced together with the other waiting lanes
//   DONE   section queue empty, lane idle
#ifndef JXL_HIP_ENTROPY_LANES_H_
#define JXL_HIP_ENTROPY_LANES_H_

#include "jxl_hip_kernels.h"
#include "jxl_hip_lanes_trip.inc"

namespace jxlhip {

struct EntropyLaneBatch {
  const EntropyParams* params;  // one per frame of the batch (device memory)
  const uint32_t* wg_unit;      // per WAVE (workgroup * WPG + wave): its unit = sections of one frame that use one histogram set
                                // (the waves of a workgroup serve the same frame, set and pass: they share its tables)
  const uint4* units;           // per unit: {index into params | histogram selector << 16, first entry in `list`, entries, pass}
  const uint32_t* list;         // group (AC section) indices of every unit, largest compressed size first
  uint32_t* queue;              // per unit: next entry of its list to hand out (zeroed before the launch). A lane takes a
                                // section, decodes it and comes back for the next one, so the lanes of a unit share its
                                // sections by actual decode time (a launch lasts as long as its busiest lane; sections
                                // differ 8x in token count, which their byte size predicts poorly)
  const uint8_t* wave_lanes;    // per wave: populated lanes (the others idle)
  uint32_t prio;                // non-zero: the waves raise their issue priority (s_setprio 3)
  uint32_t extra_pass_min;      // a service phase makes a further round of transitions only for at least this many lanes (>= 1)
  uint32_t wait_shift;          // the service phase runs once (waiting lanes << wait_shift) >= runnable lanes
  uint32_t refill_mask;         // a refill round every (refill_mask + 1) rounds (a power of two)
  const uint8_t* wave_log_ls;   // per wave: where its lanes sit in the per-wave LDS rows (64 lanes wide): bit 7 set = the
                                // workgroup's waves share ONE set of rows and this wave's lane l uses column (low 6 bits) + l
                                // (a few lanes for the frame's largest sections beside a wave for all the others, at the LDS
                                // cost of one wave); bit 7 clear = rows of its own, column l
  uint32_t debug;               // measurement aid: bit 0 = skip the coefficient stores (results are then invalid),
                                // bit 1 = report every section's coefficient-token count in its error word
  unsigned long long* started;  // optional (may be NULL): every workgroup adds 1 when it begins (the host gates other launches on
                                // "all of this launch's workgroups hold their LDS": jxl_hip_api.hip EntropyGate)
  unsigned long long* prof;     // optional (may be NULL): per wave {cycles total, cycles in service, services, hot trips,
                                // lane-trips taken, 0, 0, 0}
};

// LDS layout of one workgroup (byte offsets, every region 16-byte aligned); the host sizes the launch with it.
struct LanesLds {
  uint32_t f2, alias, ctx, ctx2, cfg, poff, wave0, per_wave, total;
};
constexpr uint32_t kLanesF2Bytes = 128;  // 2 * kCoeffFreqContext[b], b < 128, at LDS offset 0 (jxl_hip_lanes_trip.inc reads it with no base)
// Per-wave LDS, all [row][lane] with a row of 64 entries (conflict-free):
constexpr uint32_t kLanesNzRows = 96;               // nzeros line buffer [channel * 32 + column], u8
constexpr uint32_t kLanesRingWords = 16;            // stream ring, u32; + 2 mirror rows (a 3-word read at slot 15 needs no wrap)
constexpr uint32_t kLanesBlockRing = 8;             // packed block records, u32
constexpr int kLanesTrips = 4;                      // hot trips per transition pass
constexpr uint32_t kLanesRefillEvery = 4;           // rounds (trips + pass) per refill round; a power of two
constexpr uint32_t kLanesPerLaneBytes = kLanesNzRows + (kLanesRingWords + 2) * 4 + kLanesBlockRing * 4;
// alias_lds = false: the alias tables stay in global memory (k_entropy_lanes<..., GALIAS = true>); prefix = true: prefix
// codes (no alias tables at all; the per-cluster table offsets get a 1 KB region).
// a6 = true: the six-byte form of the alias tables (jxl_hip_lanes_trip.inc, variant 6: log_alpha <= 7, at most 128 clusters
// and 8192 slots; fixed regions of 16 KB + 32 KB + 256 B whatever the table size, so that the trip's offsets are immediates).
constexpr uint32_t kLanesA6Clusters = 128, kLanesA6Slots = 8192, kLanesA6B = 128, kLanesA6A = kLanesA6B + kLanesA6Slots * 2,
                   kLanesA6Cfg = kLanesA6A + kLanesA6Slots * 4, kLanesA6End = kLanesA6Cfg + kLanesA6Clusters * 2;
static_assert(kLanesA6A == 16512 && kLanesA6Cfg == 49280, "jxl_hip_lanes_trip.inc LT_EREAD_6 has these as immediates");
__host__ __device__ inline LanesLds LanesLdsLayout(uint32_t num_hist, uint32_t nctx, uint32_t num_clusters, uint32_t log_alpha,
                                                   uint32_t waves, uint32_t lanes, bool alias_lds = true, bool prefix = false, bool a6 = false) {
  LanesLds l;
  l.f2 = 0;
  l.alias = kLanesF2Bytes;  // (jxl_hip_lanes_trip.inc: ds_read_b64 ... offset:128)
  l.ctx = a6 ? kLanesA6End : l.alias + (alias_lds ? (num_clusters << log_alpha) * 8 : 0);
  l.ctx2 = l.ctx + ((num_hist * nctx + 16 + 15) & ~15u);
  l.cfg = l.ctx2 + 64 * 2;
  l.poff = l.cfg + (prefix ? 256 * 2 : 0);  // (the rANS forms carry a cluster's uint config in its alias entries)
  l.wave0 = l.poff + (prefix ? 256 * 4 : 0);
  l.per_wave = kLanesPerLaneBytes * lanes;
  l.total = l.wave0 + waves * l.per_wave;
  return l;
}

typedef uint8_t __attribute__((address_space(3))) LdsU8;
typedef uint16_t __attribute__((address_space(3))) LdsU16;
typedef uint32_t __attribute__((address_space(3))) LdsU32;
typedef uint32_t LanesU32x2 __attribute__((ext_vector_type(2)));
typedef LanesU32x2 __attribute__((address_space(3))) LdsU32x2;

// Global stores the compiler's wait-count insertion does not see (see the header comment). `base` is wave-uniform,
// `off` the lane's byte offset from it.
__device__ __forceinline__ void LaneStore64(const void* base, uint32_t off, uint32_t lo, uint32_t hi) {
  const LanesU32x2 v = {lo, hi};
  asm volatile("global_store_dwordx2 %0, %1, %2" : : "v"(off), "v"(v), "s"(base));
}
__device__ __forceinline__ void LaneStore32(const void* base, uint32_t off, uint32_t v) {
  asm volatile("global_store_dword %0, %1, %2" : : "v"(off), "v"(v), "s"(base));
}
__device__ __forceinline__ void LaneStore16(const void* base, uint32_t off, uint32_t v) {
  asm volatile("global_store_short %0, %1, %2" : : "v"(off), "v"(v), "s"(base));
}
// LDS-DMA of one dword per active lane: lane l's word at `base + off` lands at LDS byte address `lds_row + 4 * l`
// (`lds_row` wave-uniform). Counted by vmcnt like any load; the caller waits (LaneDmaWait) before it reads the row.
__device__ __forceinline__ void LaneDmaDword(uint32_t lds_row, const void* base, uint32_t off) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, %3\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(off), "s"(lds_row), "s"(base)
      : "memory");
}
__device__ __forceinline__ void LaneDmaWait() { asm volatile("s_waitcnt vmcnt(0)" : : : "memory"); }
// A 16-byte load per lane of `mask` into registers, hidden from the compiler's wait insertion like the stores: the caller
// waits (LaneLoadWait) before it reads `d`, and nothing touches `d` in between.
typedef uint32_t LanesU32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void LaneLoad128(LanesU32x4& d, unsigned long long mask, const void* base, uint32_t off) {
  asm volatile("s_mov_b64 s[96:97], exec\n\ts_and_b64 exec, exec, %3\n\tglobal_load_dwordx4 %0, %1, %2\n\ts_mov_b64 exec, s[96:97]"
               : "+v"(d)
               : "v"(off), "s"(base), "s"(mask)
               : "memory", "s96", "s97", "scc");
}
// Waits until the loads of LaneLoad128 have landed. vmcnt(0), i.e. also for every coefficient store issued since: loads
// and stores share the counter on gfx9 but do NOT complete in order with each other (a younger store may be acknowledged
// before an older load returns), so "at most n outstanding" with n > 0 says nothing about the load. (Tried: n = a lower
// bound of the stores issued since; sections then decoded garbage on and off.)
__device__ __forceinline__ void LaneLoadWait(LanesU32x4& a, LanesU32x4& b) {
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b) : : "memory");
}

// Hybrid-uint value of a token with extra bits (dec_ans.h:170-197); cfg = split_exp | msb << 4 | lsb << 8, boff = bit
// offset of the extra bits in the window w0 | w1 << 32 | w2 << 64 (< 64). Returns the value, `nb` = bits consumed.
__device__ __forceinline__ uint32_t LaneHybrid(uint32_t tok, uint32_t cfg, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t boff,
                                               uint32_t& nb) {
  const uint32_t se = cfg & 15, msb = (cfg >> 4) & 15, lsb = (cfg >> 8) & 15;
  nb = (se - (msb + lsb) + ((tok - (1u << se)) >> (msb + lsb))) & 31u;
  const uint32_t low = tok & ((1u << lsb) - 1), top = tok >> lsb;
  const bool up = boff >= 32;
  const uint32_t xb = __builtin_amdgcn_alignbit(up ? w2 : w1, up ? w1 : w0, boff & 31) & ((1u << nb) - 1);
  return (((((1u << msb) | (top & ((1u << msb) - 1))) << nb) | xb) << lsb) | low;
}

// AIDS = false compiles the measurement aids (B.prof, B.debug) out. GALIAS: the alias entries are read in place from
// global memory (tables that would not leave every frame of the launch resident in LDS). PREFIX: prefix codes
// (dec_huffman.h:28-41 in the two-level table form the host builds) instead of rANS.
// ASMT: the hot trips are the hand-written group of jxl_hip_lanes_trip.inc (LDS alias tables, rANS, int16 coefficients only);
// the C++ trip below is the statement of the algorithm, serves every other form and stays selectable for this one
// (JXLHIP_LANES_CPP=1) so that the two can be held against each other bit for bit.
// A6: the alias tables sit in LDS in the six-byte form (LanesLdsLayout): libjxl-sized tables, two frames per CU instead of one.
template <typename CoefT, int WPG, bool AIDS, bool GALIAS = false, bool PREFIX = false, bool ASMT = false, bool A6 = false>
__global__ __launch_bounds__(64 * WPG) void k_entropy_lanes(EntropyLaneBatch B_) {
  static_assert(!ASMT || (!GALIAS && !PREFIX && sizeof(CoefT) == 2), "the assembly trip: LDS alias tables, int16 coefficients");
  static_assert(!A6 || (!GALIAS && !PREFIX), "the six-byte form is an LDS form of the rANS tables");
  EntropyLaneBatch B = B_;
  if (!AIDS) {
    B.prof = nullptr;
    B.debug = 0;
  }
  extern __shared__ __align__(16) uint8_t lds_raw[];
  if (B_.prio) __builtin_amdgcn_s_setprio(3);
  const uint32_t tid = threadIdx.x, lane = tid & 63;
  if (B_.started && tid == 0) __hip_atomic_fetch_add(B_.started, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // ---- the launch description and the frame's parameter block, through the scalar cache into SGPRs
  typedef const uint32_t __attribute__((address_space(4)))* CU32;
  const uint32_t wid = blockIdx.x * WPG + wave;
  const uint32_t unit = ((CU32)(uintptr_t)B.wg_unit)[wid];
  const uint32_t wl = B.wave_log_ls[wid];
  const uint32_t lane_off = __builtin_amdgcn_readfirstlane(wl & 63), region = __builtin_amdgcn_readfirstlane((wl & 0x80) ? 0u : wave);
  const CU32 ud = (CU32)(uintptr_t)(B.units + unit);
  const uint32_t wg_desc = ud[0], list_begin = ud[1], list_count = ud[2], pass = ud[3];
  const uint32_t wg_sel = wg_desc >> 16;  // the histogram set this workgroup's sections use
  EntropyParams P;
  LoadParams(P, B.params + (wg_desc & 0xFFFF));
  PassDev T;
  LoadParams(T, P.passes + pass);
  const uint32_t log_alpha = T.log_alpha, log_entry = 12 - log_alpha, nclusters = T.num_clusters;
  const uint32_t entry_mask = (1u << log_entry) - 1, cl_shift = (A6 ? 1 : 3) + log_alpha;
  const uint32_t num_bctx = P.num_bctx, nctx = P.nctx, num_hist = P.num_hist;
  const LanesLds L = LanesLdsLayout(1, nctx, nclusters, log_alpha, 0, 0, !GALIAS && !PREFIX, PREFIX, A6);
  const uint2* const galias = T.alias_packed;
  const uint32_t* const ptable = T.prefix_table;
  LdsU8* const lds = (LdsU8*)lds_raw;
  // an alias entry from LDS in the 8-byte form whatever the layout (the C++ trip, the transition pass)
  auto alias_entry = [&](uint32_t cluster, uint32_t slot) -> LanesU32x2 {
    if constexpr (A6) {
      const uint32_t i = ((cluster << log_alpha) + slot) & (kLanesA6Slots - 1);
      const uint32_t b = ((LdsU16*)(lds + kLanesA6B))[i], a = ((LdsU32*)(lds + kLanesA6A))[i];
      const uint32_t cfg = ((LdsU16*)(lds + kLanesA6Cfg))[cluster & (kLanesA6Clusters - 1)];
      const uint32_t cutoff = (b >> 12) << (8 - log_alpha) | a >> (24 + log_alpha);
      return LanesU32x2{(b & 0xFFFu) | cfg << 12 | cutoff << 24, a & ((1u << (24 + log_alpha)) - 1)};
    } else {
      return *(LdsU32x2*)((LdsU8*)lds + kLanesF2Bytes + (cluster << cl_shift) + slot * 8);
    }
  };
  LdsU8* const l_ctx = lds + L.ctx;                      // context -> histogram (cluster)
  LdsU16* const l_nnz2 = (LdsU16*)(lds + L.ctx2);        // [ceil(nzeros left / covered)] -> 2 * kCoeffNumNonzeroContext
  LdsU16* const l_cfg = (LdsU16*)(lds + L.cfg);          // per cluster: split_exp | msb << 4 | lsb << 8
  LdsU32* const l_poff = (LdsU32*)(lds + L.poff);        // (PREFIX only)
  const uint32_t wave_base = L.wave0 + region * (kLanesPerLaneBytes * 64);
  const uint32_t col = lane_off + lane;  // this lane's column of the [row][64] arrays (lanes beyond the populated ones only read)
  LdsU8* const l_nz = lds + wave_base + col;                                          // nzeros line buffer [row][lane]
  const uint32_t ring_row0 = wave_base + kLanesNzRows * 64;                           // byte offset of stream-ring row 0
  const uint32_t bring_row0 = ring_row0 + (kLanesRingWords + 2) * 256;                // ... of block-record row 0
  LdsU32* const ring = (LdsU32*)(lds + ring_row0) + col;
  LdsU32* const bring = (LdsU32*)(lds + bring_row0) + col;
  // LDS address of the dynamic segment + this wave's column offset (the DMA wants absolute addresses: row + 4 * lane)
  const uint32_t lds_abs = uint32_t(uintptr_t(lds)) + lane_off * 4;

  // ---- stage the frame's tables (whole workgroup)
  {
    const uint32_t n_ctx = nctx + 16;  // the selected set's slice of the context map
    const uint8_t* ctx_slice = T.ctx_map + size_t(wg_sel) * nctx;
    for (uint32_t i = tid; i < n_ctx; i += 64 * WPG) {
      const uint32_t cl = ctx_slice[i];
      l_ctx[i] = uint8_t(cl < nclusters ? cl : nclusters - 1);
    }
    if (PREFIX)
      for (uint32_t i = tid; i < nclusters; i += 64 * WPG) {
        const uint32_t cfg = T.cfg[i];
        l_cfg[i] = uint16_t((cfg & 15) | (((cfg >> 8) & 15) << 4) | (((cfg >> 16) & 15) << 8));
      }
    // alias entry {cutoff u8, right u8, freq0 u16 | offsets1 u16, freq1 u16} ->
    //   x = (freq0 - 1) & 0xFFF | uint config << 12 | cutoff << 24   taken when pos <  cutoff: symbol = slot, offset = pos
    //   y = (freq1 - 1) & 0xFFF | offsets1 << 12 | right << 24        taken when pos >= cutoff
    // (repacked by jxlhip_frame_upload: PassDev::alias_packed; the uint config of the cluster rides in the spare bits
    // of x, so the trip needs no separate lookup for it)
    if (PREFIX)
      for (uint32_t i = tid; i < nclusters; i += 64 * WPG) l_poff[i] = T.prefix_offset[i];
    const uint32_t n_alias = (GALIAS || PREFIX) ? 0u : nclusters << log_alpha;
    if constexpr (A6) {
      // the six-byte form (the host picks it for log_alpha <= 7, at most kLanesA6Clusters and kLanesA6Slots only): a cutoff of
      // 2^(12 - log_alpha) (every position of the slot is its own symbol) becomes one less, with the slot's own symbol and
      // frequency on the other side at offset 0
      LdsU16* const lb = (LdsU16*)(lds + kLanesA6B);
      LdsU32* const la = (LdsU32*)(lds + kLanesA6A);
      LdsU16* const lc = (LdsU16*)(lds + kLanesA6Cfg);
      for (uint32_t i = tid; i < n_alias; i += 64 * WPG) {
        const uint2 e = galias[i];
        uint32_t cutoff = e.x >> 24, y = e.y & ((1u << (24 + log_alpha)) - 1);
        if (cutoff > entry_mask) {
          cutoff = entry_mask;
          y = (e.x & 0xFFFu) | (i & ((1u << log_alpha) - 1)) << 24;
        }
        const uint32_t lo_bits = 8 - log_alpha;
        lb[i] = uint16_t((e.x & 0xFFFu) | (cutoff >> lo_bits) << 12);
        la[i] = y | (cutoff & ((1u << lo_bits) - 1)) << (24 + log_alpha);
        if ((i & ((1u << log_alpha) - 1)) == 0) lc[i >> log_alpha] = uint16_t((e.x >> 12) & 0xFFFu);
      }
    } else {
      LdsU32x2* l_alias = (LdsU32x2*)(lds + L.alias);
      for (uint32_t i = tid; i < n_alias; i += 64 * WPG) {
        const uint2 e = galias[i];
        l_alias[i] = LanesU32x2{e.x, e.y};
      }
    }
    if (tid < 64) l_nnz2[tid] = uint16_t(uint32_t(c_coeff_nnz_ctx[tid]) * 2);
    // 2 * kCoeffFreqContext(b) (ac_context.h:63-80), b = (k + 1) >> log2 covered: 1..63 in a valid stream; the entries
    // beyond continue the closed form (a corrupt stream may run a few positions past its block before it is stopped)
    for (uint32_t b = tid; b < kLanesF2Bytes; b += 64 * WPG)
      lds[L.f2 + b] = uint8_t(b < 64 ? uint32_t(c_coeff_freq_ctx[b]) * 2 : 2 * min(min(b - 1, 7 + (b >> 1)), 15 + (b >> 2)));
    LdsU32* z = (LdsU32*)(lds + wave_base);
    for (uint32_t i = lane; i < kLanesNzRows * 64 / 4; i += 64) z[i] = 0;
  }
  __syncthreads();

  // ---- per-lane section setup
  enum : uint32_t { kWait = 0, kRun = 1, kFlush = 2, kDone = 3 };
  auto take_section = [&]() -> uint32_t {
    const uint32_t idx = atomicAdd(B.queue + unit, 1u);
    return idx < list_count ? B.list[list_begin + idx] : 0xFFFFFFFFu;
  };
  uint32_t g = lane < B.wave_lanes[wid] ? take_section() : 0xFFFFFFFFu;
  uint32_t mode = g == 0xFFFFFFFFu ? uint32_t(kDone) : uint32_t(kWait);
  uint32_t err = 0, ntok = 0;  // ntok: measurement aid (debug bit 1: the flag word reports the section's token count)
  uint32_t b1 = 0, bi = 0, ci = 2;
  const uint32_t sec0 = pass * P.num_groups;  // the pass's first entry in the section tables
  const void* const kend_base = P.kend + size_t(pass) * P.kend_pass_stride;
  const void* const coef_base = static_cast<CoefT*>(P.coeffs) + P.coef_pass_base + size_t(pass) * P.coef_pass_stride;
  // stream cursor: `ring_end` words of the section have landed in the ring, `pend_s` more are in flight (DMA)
  uint32_t sec_off = 0, nwords = 0, sec_size = 0, ring_end = 0, pend_s = 0, bring_end = 0, pend_b = 0, bitpos = 0, state = 0;
  LanesU32x4 stg = {0, 0, 0, 0}, btg = {0, 0, 0, 0};  // (ASMT) stream words / block records on their way into the rings
  uint32_t stg_n = 0, btg_n = 0;  // 4: requested; 8: requested past the section's end (zeros)
  bool started = false;
  // block / channel cursor
  uint32_t info = 0, lbx = 0, lby = 0, coef_offset = 0, next_offset = 0;
  auto open_section = [&]() {  // cursors of section g (the nzeros line buffer needs no reset: every entry a section
                               // reads was written by an earlier block of the same section)
    const uint32_t first = P.gbb[g];
    bi = first - 1;            // the first transition advances to the group's first block
    b1 = P.gbb[g + 1];
    bring_end = first;
    sec_off = P.sec_word[sec0 + g] * 4;  // 16-byte aligned (jxlhip_frame_upload)
    sec_size = P.sec_size[sec0 + g];
    nwords = (sec_size + 3) / 4;
    bitpos = sec0 + g == 0 ? P.first_bit_offset : 0;
    ring_end = 0;
    pend_s = pend_b = 0;       // (what is still in flight for the finished section lands before the next request is made)
    stg_n = btg_n = 0;
    next_offset = 0;
    ci = 2;
    err = 0;
    started = false;
  };
  if (mode == kWait) open_section();
  // coefficient cursor
  // addr_a / nnz_b: LDS byte address (relative to the context map) of the context entry of the NEXT coefficient at
  // frequency context 0, if the current token turns out zero (same non-zero count, prev = 0) / the raw table value from
  // which the address for a non-zero token (one fewer to come, prev = 1) is formed where it is used
  // (ASMT: addr_a and cbase are LDS byte addresses, i.e. include the context map's offset; covm2 = covm1 - 2)
  uint32_t nzeros = 0, k = 0, size = 0, log2c = 0, covm1 = 0, cbase = 0, addr_a = 0, nnz_b = 0, ctxe = 0, kidx = 0, covm2 = 0;
  const uint32_t ring_addr = uint32_t(uintptr_t(ring)), cl_mul = 1u << cl_shift, nnz2_addr = uint32_t(uintptr_t(l_nnz2));
  if (ASMT && uint32_t(uintptr_t(lds)) != 0) __builtin_trap();  // (jxl_hip_lanes_trip.inc addresses the tables from LDS offset 0)
  uint32_t acc_lo = 0, acc_hi = 0;  // the coefficient chunk in progress
  uint32_t dst = 0;                 // byte offset (from coef_base) of the chunk in progress
  const uint32_t shift = T.shift;
  constexpr uint32_t kPerChunk = 8 / sizeof(CoefT);

  unsigned long long t_begin = 0, t_service = 0, n_service = 0, n_trips = 0, n_lane_trips = 0, t_wait = 0, t_dma = 0, t_trans = 0, n_calls = 0;
  if (B.prof) t_begin = __builtin_readcyclecounter();
  __builtin_amdgcn_s_waitcnt(0);  // everything loaded so far has landed: the loop starts with empty counters
  // One 4-row group of a ring: the rows' LDS-DMA requests for the lanes in `in` (word index `w0i` at the group's first
  // row), in one statement so that M0 is saved and restored once.
  auto dma_group4 = [&](bool in, uint32_t row_lds, const void* base, uint32_t off) {
    if (in) {
      // (an instruction offset would move the LDS address as well as the global one: the rows' global offsets are
      // formed in a scratch register)
      uint32_t keep, tmp;
      asm volatile(
          "s_mov_b32 %0, m0\n\t"
          "s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %2, %4\n\t"
          "v_add_u32 %1, 4, %2\n\ts_add_u32 m0, m0, 0x100\n\ts_nop 0\n\tglobal_load_lds_dword %1, %4\n\t"
          "v_add_u32 %1, 8, %2\n\ts_add_u32 m0, m0, 0x100\n\ts_nop 0\n\tglobal_load_lds_dword %1, %4\n\t"
          "v_add_u32 %1, 12, %2\n\ts_add_u32 m0, m0, 0x100\n\ts_nop 0\n\tglobal_load_lds_dword %1, %4\n\t"
          "s_mov_b32 m0, %0"
          : "=&s"(keep), "=&v"(tmp)
          : "v"(off), "s"(row_lds), "s"(base)
          : "memory", "scc");
    }
  };
  // The loop: kLanesTrips hot trips, then ONE transition pass, each a straight line of code for every lane that is in
  // the matching state; every kLanesRefillEvery-th round also the ring refills and the rare transitions (section header,
  // section complete). A lone wave pays per instruction (~6 cycles with its LDS waits: rocprofv3 SQ counters,
  // profiles/r03_*), and far more per taken branch, so nothing here waits for "enough" lanes: a pass costs about one
  // trip, a lane waits at most kLanesTrips trips for it.
  uint32_t round = 0;
  for (;;) {
    // ================================================================= hot trips: one coefficient token per lane each
    // A trip is ONE basic block: every lane computes, the lanes that are not taking part keep their state through
    // selects. Lanes outside the trip read LDS at stale addresses: harmless (an LDS address outside the allocation reads
    // as zero). Only the coefficient store is predicated.
    {
      // a group of kLanesTrips hot trips consumes at most kLanesTrips * 47 bits (16 renormalisation + 31 extra bits each),
      // i.e. bits of at most 2 * kLanesTrips ring words (words read beyond ring_end are stale but never consumed)
      const bool low = (ring_end - (bitpos >> 5)) < 2 * kLanesTrips + 1;
      bool act = mode == kRun && !low;
      if constexpr (ASMT) {
        const unsigned long long actm = __ballot(act);
        if (actm) {
          unsigned long long th = 0;
          const uint32_t k_before = k;
          // The loop goes on, group after group, until the transition pass is due: as many lanes waiting for it as the pass
          // threshold asks for ((waiting << wait_shift) >= running, the rule of the C++ form), counting the lanes that wait
          // already and those that leave the loop.
          const uint32_t n_total = uint32_t(__popcll(actm)) + uint32_t(__popcll(__ballot(mode == kWait || mode == kFlush)));
          const uint32_t cont_min = __builtin_amdgcn_readfirstlane(((n_total << B.wait_shift) / (1u + (1u << B.wait_shift))) + 1u);
          uint32_t groups = 0;
          // (operands: the lane's decoder state and cursors; its per-run constants; the loop's lanes and threshold; the frame's constants)
          if (AIDS) {
            if (B.prof) th = __builtin_readcyclecounter();
            if constexpr (A6) {
              JXL_LANES_TRIP_LOOP6_COUNTED(groups, state, bitpos, k, nzeros, ctxe, addr_a, nnz_b, dst, acc_lo, acc_hi, log2c, cbase, covm2, ring_addr,
                                           ring_end, size, actm, cont_min, log_entry, entry_mask, cl_mul, nnz2_addr, shift, coef_base, log_alpha);
            } else {
              JXL_LANES_TRIP_LOOP_COUNTED(groups, state, bitpos, k, nzeros, ctxe, addr_a, nnz_b, dst, acc_lo, acc_hi, log2c, cbase, covm2, ring_addr,
                                          ring_end, size, actm, cont_min, log_entry, entry_mask, cl_mul, nnz2_addr, shift, coef_base, log_alpha);
            }
            if (B.prof) t_wait += __builtin_readcyclecounter() - th;
            n_trips += groups * kLanesTrips;
            n_calls++;
          } else {
            if constexpr (A6) {
              JXL_LANES_TRIP_LOOP6(groups, state, bitpos, k, nzeros, ctxe, addr_a, nnz_b, dst, acc_lo, acc_hi, log2c, cbase, covm2, ring_addr, ring_end,
                                   size, actm, cont_min, log_entry, entry_mask, cl_mul, nnz2_addr, shift, coef_base, log_alpha);
            } else {
              JXL_LANES_TRIP_LOOP(groups, state, bitpos, k, nzeros, ctxe, addr_a, nnz_b, dst, acc_lo, acc_hi, log2c, cbase, covm2, ring_addr, ring_end,
                                  size, actm, cont_min, log_entry, entry_mask, cl_mul, nnz2_addr, shift, coef_base, log_alpha);
            }
          }
          // lanes of the group that finished their (block, channel), or ran past its last scan position (corrupt stream:
          // the transition pass sees non-zeros left and abandons the section)
          mode = (act && (nzeros == 0 || k >= size)) ? uint32_t(kFlush) : mode;
          if (AIDS && B.prof) n_lane_trips += act ? k - k_before : 0u;
          if (AIDS && (B.debug & 2)) ntok += act ? k - k_before : 0u;
        }
      } else
      if (__ballot(act)) {
        unsigned long long th = 0;
        if (B.prof) th = __builtin_readcyclecounter();
        if (AIDS) n_trips += kLanesTrips;
        // Software pipeline (the LDS-alias rANS form): what a trip reads from LDS is requested by the trip BEFORE it, as soon
        // as that trip knows its new state, bit position and context, ahead of its own bookkeeping (coefficient chunk,
        // store, end-of-channel test): a lone wave has nothing else to cover the LDS round trip with. The group's first trip
        // requests its own reads here.
        // (GALIAS: the same pipeline with the alias entry as a global load: one trip's worth of instructions between the
        // request and the use covers most of an L2 hit. PREFIX: the root lookup of the next token's code leaves as soon as
        // the window words and the cluster's table offset are back from LDS; a second-level lookup, for codes longer than
        // the root index, stays inside the trip.)
        uint32_t pw0 = 0, pw1 = 0, pw2 = 0, pe_zero = 0, pe_nonzero = 0, pnnz_c = 0;
        uint32_t ppo = 0, pcfg = 0, proot = 0;  // (PREFIX)
        LanesU32x2 pe = {0, 0};
        auto request = [&]() {
          LdsU32* const rp = ring + ((bitpos >> 5) & (kLanesRingWords - 1)) * 64;
          if (PREFIX) {
            ppo = l_poff[ctxe & 255];
            pcfg = l_cfg[ctxe & 255];
          } else if (GALIAS) {
            if (act) {  // (only lanes of the trip: a lane's entry is a cache line of its own to the memory pipeline)
              const uint2 ge = galias[(ctxe << log_alpha) + ((state & 0xFFFu) >> log_entry)];
              pe = LanesU32x2{ge.x, ge.y};
            }
          } else {
            pe = alias_entry(ctxe, (state & 0xFFFu) >> log_entry);
          }
          pw0 = rp[0];
          pw1 = rp[64];
          pw2 = rp[128];
          __builtin_amdgcn_sched_barrier(0);  // (the reads on the serial chain leave first; the address arithmetic below covers them)
          // the context entries of coefficient k + 1 for both outcomes of coefficient k (off the serial chain);
          // kCoeffFreqContext(b) for b = (k + 1) / covered in 1..63 is min(b - 1, 7 + b / 2, 15 + b / 4)
          const uint32_t b = (k + 1) >> log2c;
          const uint32_t f2 = min(min(b - 1, 7 + (b >> 1)), 15 + (b >> 2)) << 1;
          pe_zero = l_ctx[(GALIAS || PREFIX) ? ((addr_a + f2) & 0x1FFF) : (addr_a + f2)];
          pe_nonzero = l_ctx[(GALIAS || PREFIX) ? ((cbase + 1 + nnz_b + f2) & 0x1FFF) : (cbase + 1 + nnz_b + f2)];
          // nnz table entry for the trip after next if this token is non-zero (if it is zero, nnz_b stays)
          pnnz_c = l_nnz2[((nzeros - 2 + covm1) >> log2c) & 63];
          if (PREFIX) {
            const uint32_t win = __builtin_amdgcn_alignbit(pw1, pw0, bitpos);
            proot = 0;
            if (act) proot = ptable[(ppo & 0xFFFFFFu) + (win & ((1u << (ppo >> 24)) - 1))];  // (global table: only from valid state)
          }
        };
        request();
#pragma unroll
        for (int rep = 0; rep < kLanesTrips; rep++) {
          if (AIDS && B.prof) n_lane_trips += act ? 1u : 0u;
          uint32_t w0, w1, w2, e_zero, e_nonzero, nnz_c;
          const uint32_t kn = k + 1;
          const uint32_t addr_b = cbase + 1 + nnz_b;
          uint32_t tok, cfg, adv, nstate = state;
          w0 = pw0, w1 = pw1, w2 = pw2, e_zero = pe_zero, e_nonzero = pe_nonzero, nnz_c = pnnz_c;
          if (PREFIX) {
            uint32_t e = proot;
            if (act && (e & 0x80u)) {  // second level (jxl_hip_kernels.h PrefixLookup)
              const uint32_t win = __builtin_amdgcn_alignbit(w1, w0, bitpos);
              e = ptable[(ppo & 0xFFFFFFu) + (e >> 8) + ((win >> (ppo >> 24)) & ((1u << (e & 0x7Fu)) - 1))];
            }
            tok = e >> 8;
            adv = e & 0xFFu;
            cfg = pcfg;
          } else {
            const LanesU32x2 e = pe;
            const uint32_t res = state & 0xFFFu, slot = res >> log_entry, pos = res & entry_mask;
            const bool gt = pos >= (e.x >> 24);
            const uint32_t x = gt ? e.y : e.x;
            tok = gt ? (e.y >> 24) : slot;
            const uint32_t hi = state >> 12;
            nstate = (x & 0xFFFu) * hi + hi + (gt ? ((e.y >> 12) & 0xFFFu) : 0u) + pos;
            const bool need = nstate < (1u << 16);
            const uint32_t win = __builtin_amdgcn_alignbit(w1, w0, bitpos);
            nstate = need ? ((nstate << 16) | (win & 0xFFFFu)) : nstate;
            adv = need ? 16u : 0u;
            cfg = e.x >> 12;
          }
          // Tokens with extra bits are rare (|coefficient| >= 8 at the usual split of 16): their ~25 instructions are
          // skipped when no lane of the trip has one (a wave-uniform branch).
          const bool take = act && tok >= (1u << (cfg & 15));
          if (__ballot(take)) {
            uint32_t nbits;
            const uint32_t big = LaneHybrid(tok, cfg, w0, w1, w2, (bitpos & 31) + adv, nbits);
            tok = take ? big : tok;
            adv += take ? nbits : 0u;
          }
          const bool nz = tok != 0;
          const uint32_t n_nzeros = nzeros - (nz ? 1u : 0u);
          const bool done = n_nzeros == 0 || kn >= size;
          const bool full = act && (kn & (kPerChunk - 1)) == 0;
          // ---- commit what the next trip's reads depend on (lanes outside the trip keep everything), then request them
          const uint32_t acc_lo0 = acc_lo, acc_hi0 = acc_hi, dst0 = dst;
          state = act ? nstate : state;
          bitpos += act ? adv : 0u;
          k = act ? kn : k;
          nzeros = act ? n_nzeros : nzeros;
          ctxe = act ? (nz ? e_nonzero : e_zero) : ctxe;
          addr_a = (act && nz) ? addr_b - 1 : addr_a;
          nnz_b = (act && nz) ? nnz_c : nnz_b;
          if (rep + 1 < kLanesTrips) request();
          // ---- the trip's own bookkeeping, under the round trip of those reads
          const uint32_t sgn = uint32_t(-int32_t(tok & 1));  // odd token: negative
          const int32_t coeff = int32_t(((tok >> 1) ^ sgn) << shift);
          // Coefficients leave in aligned 8-byte chunks (4 x int16 / 2 x int32), shifted into a register pair from the top:
          // a lane's stores go to 64 different cache lines, and the memory pipeline takes such a wave-instruction one lane
          // at a time (scripts/ubench_trip.hip). The chunk of scan positions [4j, 4j + 4) is stored when position 4j + 3
          // has been decoded; the transition pass stores the last, partial one. Entries outside [covered, kend) of a chunk
          // are unspecified (the transforms never read them).
          uint32_t n_lo, n_hi;
          if (sizeof(CoefT) == 2) {
            n_lo = __builtin_amdgcn_alignbit(acc_hi0, acc_lo0, 16);
            n_hi = (acc_hi0 >> 16) | (uint32_t(coeff) << 16);
          } else {
            n_lo = acc_hi0;
            n_hi = uint32_t(coeff);
          }
          if ((B.debug & 1) == 0 && full) LaneStore64(coef_base, dst0, n_lo, n_hi);
          if (AIDS && (B.debug & 2)) ntok += act ? 1u : 0u;
          acc_lo = act ? n_lo : acc_lo0;
          acc_hi = act ? n_hi : acc_hi0;
          dst = dst0 + (full ? 8u : 0u);
          mode = (act && done) ? uint32_t(kFlush) : mode;
          act = act && !done;
        }
        if (B.prof) t_wait += __builtin_readcyclecounter() - th;  // (cycles in hot trips)
      }
    }
    // ================================================================= transition pass: the next (block, channel) of
    // every lane that finished one (its last coefficient chunk and kend first) or is still between two, including the
    // (block, channel)'s non-zero-count symbol. A channel that turns out empty leaves its lane waiting for the next pass.
    {
      unsigned long long t2 = 0;
      if (B.prof) t2 = __builtin_readcyclecounter();
      const bool fl = mode == kFlush;
      const bool next_block = ci == 2 && started;  // the coming transition moves on to block bi + 1
      // (a lane that ran out of scan positions abandons its section: the refill round takes it from there)
      const bool ran_out = fl && nzeros != 0;
      const bool rare = !started || (next_block && bi + 1 >= b1) || ran_out;  // section header / section complete: refill round
      const bool go = (mode == kWait || fl) && !rare && (ring_end - (bitpos >> 5)) >= 3 && (!next_block || bi + 1 < bring_end);
      // (a pass costs about two trips: it runs once enough lanes wait for it, or nothing else can run)
      const uint32_t n_go = uint32_t(__popcll(__ballot(fl || go))), n_run = uint32_t(__popcll(__ballot(mode == kRun)));
      if (n_go && (n_go << B.wait_shift) >= n_run) {
        if (ASMT) {
          if (fl) {  // the 1 - 3 coefficients decoded since the lane's last whole group of four (a run starts at position
                     // `covered` and every whole group stored four): exactly those, from acc_lo | acc_hi << 32 upwards
            const uint32_t part = (k - covm1 - 1) & 3;
            if (!(B.debug & 1)) {
              if (part >= 2) LaneStore32(coef_base, dst, acc_lo);
              if (part & 1) LaneStore16(coef_base, dst + (part & 2) * 2, part == 3 ? acc_hi : acc_lo);
            }
            if (!(B.debug & 4)) LaneStore32(kend_base, kidx * 4, k);
          }
        } else
        if (fl) {  // the chunk in progress (its entries sit at the top of the register pair), the number of scan positions
          const uint32_t part = k & (kPerChunk - 1);
          const uint64_t v = ((uint64_t(acc_hi) << 32) | acc_lo) >> (((kPerChunk - part) & (kPerChunk - 1)) * (64 / kPerChunk));
          if (!(B.debug & 1) && part) LaneStore64(coef_base, dst, uint32_t(v), uint32_t(v >> 32));
          if (!(B.debug & 4)) LaneStore32(kend_base, kidx * 4, k);
        }
        err |= ran_out ? kErrNzeros : 0u;
        bi = ran_out ? b1 : bi;
        ci = ran_out ? 2u : ci;
        mode = fl ? uint32_t(kWait) : mode;
        if constexpr (ASMT) {
          // ---- the transition proper, for the `go` lanes only (one EXEC region, plain assignments: the select form below
          // costs 339 instructions per pass, a third of them scalar mask bookkeeping)
          if (go) {
            const bool nb = ci == 2;
            const uint32_t n_ci = nb ? 0u : ci + 1, n_bi = bi + (nb ? 1u : 0u);
            if (nb) {
              info = bring[(n_bi & (kLanesBlockRing - 1)) * 64];
              coef_offset = next_offset;
            }
            const uint32_t n_log2c = ((info >> 6) & 7) + ((info >> 9) & 7);
            if (nb) next_offset += 64u << n_log2c;
            const uint32_t c = n_ci == 0 ? 1u : (n_ci == 1 ? 0u : 2u);
            // (chroma-subsampled frames: record bit 24 + c = the block carries nothing for channel c, bit 27 + c = the
            // channel's columns are half the frame's; both 0 otherwise)
            if ((info >> (24 + c)) & 1u) {
              if (!(B.debug & 4)) LaneStore32(kend_base, (n_bi * 3 + c) * 4, 0);  // stays waiting: next channel / block
              bi = n_bi;
              ci = n_ci;
            } else {
            const uint32_t n_lbx = (info & 31) >> ((info >> (27 + c)) & 1u), n_lby = (info >> 5) & 1;
            const uint32_t log2cx = (info >> 6) & 7;
            const uint32_t bctx = (info >> (12 + 4 * c)) & 15;
            LdsU8* line = l_nz + (c * 32) * 64;
            const uint32_t top = line[n_lbx * 64], left = line[(n_lbx ? n_lbx - 1 : 0) * 64];
            const uint32_t l0 = n_lbx ? left : 32u, tt = n_lby ? top : l0, ll = n_lbx ? left : tt;
            const uint32_t pred = (tt + ll + 1) >> 1;
            uint32_t nzb = pred >= 64 ? 64 : pred;
            nzb = nzb < 8 ? nzb : 4 + nzb / 2;
            const uint32_t cluster = l_ctx[(nzb * num_bctx + bctx) & 0x1FFF];
            const uint32_t slotw = (bitpos >> 5) & (kLanesRingWords - 1);
            const uint32_t w0 = ring[slotw * 64], w1 = ring[slotw * 64 + 64], w2 = ring[slotw * 64 + 128];
            const uint32_t res = state & 0xFFFu, slot = res >> log_entry, pos = res & entry_mask;
            const LanesU32x2 e = alias_entry(cluster, slot);
            const bool gt = pos >= (e.x >> 24);
            const uint32_t x = gt ? e.y : e.x;
            uint32_t tok = gt ? (e.y >> 24) : slot;
            const uint32_t hi = state >> 12;
            uint32_t nstate = (x & 0xFFFu) * hi + hi + (gt ? ((e.y >> 12) & 0xFFFu) : 0u) + pos;
            const bool need = nstate < (1u << 16);
            const uint32_t win = __builtin_amdgcn_alignbit(w1, w0, bitpos);
            state = need ? ((nstate << 16) | (win & 0xFFFFu)) : nstate;
            uint32_t adv = need ? 16u : 0u;
            const uint32_t cfg = e.x >> 12;
            const bool take = tok >= (1u << (cfg & 15));
            if (__ballot(take)) {
              uint32_t nbits;
              const uint32_t big = LaneHybrid(tok, cfg, w0, w1, w2, (bitpos & 31) + adv, nbits);
              tok = take ? big : tok;
              adv += take ? nbits : 0u;
            }
            bitpos += adv;
            const uint32_t covered = 1u << n_log2c;
            log2c = n_log2c;
            size = covered * 64;
            kidx = n_bi * 3 + c;
            const bool bad = tok > size - covered;
            const uint8_t nzv = uint8_t((tok + covered - 1) >> n_log2c);
            if (!bad) {
              line[n_lbx * 64] = nzv;
              if (__ballot(log2cx != 0)) {
                const uint32_t cx = 1u << log2cx;
                for (uint32_t i = 1; i < cx; i++) line[(n_lbx + i) * 64] = nzv;
              }
            }
            if (tok == 0 && !(B.debug & 4)) LaneStore32(kend_base, kidx * 4, 0);  // stays waiting: next channel / block
            // the coefficient run's cursor (unused when the channel is empty or the count invalid)
            covm1 = covered - 1;
            covm2 = covered - 3;
            const uint32_t n_cbase = num_bctx * 37 + 458 * bctx;
            const uint32_t prev = tok > size / 16 ? 0 : 1;
            const uint32_t n_addr_a = n_cbase + l_nnz2[((tok + covm1) >> n_log2c) & 63];
            nnz_b = l_nnz2[((tok - 1 + covm1) >> n_log2c) & 63];
            ctxe = l_ctx[(n_addr_a + prev) & 0x1FFF];  // frequency context of k = covered is 0
            cbase = n_cbase + L.ctx;
            addr_a = n_addr_a + L.ctx;
            dst = uint32_t(((g * 3 + c) * 65536 + coef_offset + covered) * sizeof(CoefT));  // position `covered` itself
            nzeros = tok;
            k = covered;
            mode = (tok != 0 && !bad) ? uint32_t(kRun) : mode;
            err |= bad ? kErrNzeros : 0u;  // abandon the section: the refill round takes the "section complete" path
            bi = bad ? b1 : n_bi;
            ci = bad ? 2u : n_ci;
            }
          }
        } else {
        // ---- the transition proper (every lane computes; `go` lanes commit)
        const bool nb = ci == 2;
        const uint32_t n_ci = nb ? 0u : ci + 1, n_bi = bi + (nb ? 1u : 0u);
        const uint32_t rec = bring[(n_bi & (kLanesBlockRing - 1)) * 64];
        // packed record (jxlhip_frame_upload): column | not first row << 5 | log2 covered_x << 6 | log2 covered_y << 9 |
        // block context of X, Y, B << 12, 16, 20
        const uint32_t n_info = nb ? rec : info;
        const uint32_t n_log2c = ((n_info >> 6) & 7) + ((n_info >> 9) & 7);
        const uint32_t n_coef_offset = nb ? next_offset : coef_offset;  // blocks of a group are contiguous in its planes
        const uint32_t c = n_ci == 0 ? 1u : (n_ci == 1 ? 0u : 2u);
        // chroma-subsampled frames: record bit 24 + c = the block carries nothing for channel c (the lane moves on without
        // reading a symbol), bit 27 + c = the channel's columns are half the frame's; both 0 otherwise
        const bool absent = ((n_info >> (24 + c)) & 1u) != 0;
        const bool go_all = go;        // the block / channel cursor advances
        const bool go = go_all && !absent;  // ... and a symbol is read
        const uint32_t n_lbx = (n_info & 31) >> ((n_info >> (27 + c)) & 1u), n_lby = (n_info >> 5) & 1;
        const uint32_t log2cx = (n_info >> 6) & 7;
        const uint32_t bctx = (n_info >> (12 + 4 * c)) & 15;
        LdsU8* line = l_nz + (c * 32) * 64;
        const uint32_t top = line[n_lbx * 64], left = line[(n_lbx ? n_lbx - 1 : 0) * 64];
        // prediction (dec_group.cc:430-450 PredictFromTopAndLeft): 32 / top / left / their rounded mean
        const uint32_t l0 = n_lbx ? left : 32u, tt = n_lby ? top : l0, ll = n_lbx ? left : tt;
        const uint32_t pred = (tt + ll + 1) >> 1;
        uint32_t nzb = pred >= 64 ? 64 : pred;
        nzb = nzb < 8 ? nzb : 4 + nzb / 2;
        const uint32_t cluster = l_ctx[(nzb * num_bctx + bctx) & 0x1FFF];
        // ---- one symbol + hybrid-uint extra bits (the same arithmetic as the hot trip)
        const uint32_t slotw = (bitpos >> 5) & (kLanesRingWords - 1);
        const uint32_t w0 = ring[slotw * 64], w1 = ring[slotw * 64 + 64], w2 = ring[slotw * 64 + 128];
        uint32_t tok, cfg, adv, nstate = state;
        if (PREFIX) {
          const uint32_t po = l_poff[cluster];
          const uint32_t win = __builtin_amdgcn_alignbit(w1, w0, bitpos);
          uint32_t e = 0;
          if (go) e = PrefixLookup(ptable + (po & 0xFFFFFFu), po >> 24, win);
          tok = e >> 8;
          adv = e & 0xFFu;
          cfg = l_cfg[cluster];
        } else {
          const uint32_t res = state & 0xFFFu, slot = res >> log_entry, pos = res & entry_mask;
          LanesU32x2 e = {0, 0};
          if (GALIAS) {
            if (go) {
              const uint2 ge = galias[(cluster << log_alpha) + slot];
              e = LanesU32x2{ge.x, ge.y};
            }
          } else {
            e = alias_entry(cluster, slot);
          }
          const bool gt = pos >= (e.x >> 24);
          const uint32_t x = gt ? e.y : e.x;
          tok = gt ? (e.y >> 24) : slot;
          const uint32_t hi = state >> 12;
          nstate = (x & 0xFFFu) * hi + hi + (gt ? ((e.y >> 12) & 0xFFFu) : 0u) + pos;
          const bool need = nstate < (1u << 16);
          const uint32_t win = __builtin_amdgcn_alignbit(w1, w0, bitpos);
          nstate = need ? ((nstate << 16) | (win & 0xFFFFu)) : nstate;
          adv = need ? 16u : 0u;
          cfg = e.x >> 12;
        }
        const bool take = go && tok >= (1u << (cfg & 15));
        if (__ballot(take)) {
          uint32_t nbits;
          const uint32_t big = LaneHybrid(tok, cfg, w0, w1, w2, (bitpos & 31) + adv, nbits);
          tok = take ? big : tok;
          adv += take ? nbits : 0u;
        }
        // ----
        const uint32_t covered = 1u << n_log2c, n_size = covered * 64;
        const uint32_t n_kidx = n_bi * 3 + c;
        const bool bad = tok > n_size - covered;
        const uint8_t nzv = uint8_t((tok + covered - 1) >> n_log2c);
        if (go && !bad) {
          line[n_lbx * 64] = nzv;
          if (__ballot(log2cx != 0)) {
            const uint32_t cx = 1u << log2cx;
            for (uint32_t i = 1; i < cx; i++) line[(n_lbx + i) * 64] = nzv;
          }
        }
        if (((go && tok == 0) || (go_all && absent)) && !(B.debug & 4)) LaneStore32(kend_base, n_kidx * 4, 0);  // stays waiting: next channel / block
        // the coefficient run's cursor (unused when the channel is empty or the count invalid)
        const uint32_t n_covm1 = covered - 1;
        const uint32_t n_cbase = num_bctx * 37 + 458 * bctx;
        const uint32_t prev = tok > n_size / 16 ? 0 : 1;
        const uint32_t n_addr_a = n_cbase + l_nnz2[((tok + n_covm1) >> n_log2c) & 63];
        const uint32_t n_nnz_b = l_nnz2[((tok - 1 + n_covm1) >> n_log2c) & 63];
        const uint32_t n_ctxe = l_ctx[(n_addr_a + prev) & 0x1FFF];  // frequency context of k = covered is 0
        // ---- commit
        state = go ? nstate : state;
        bitpos += go ? adv : 0u;
        info = go_all ? n_info : info;
        coef_offset = go_all ? n_coef_offset : coef_offset;
        next_offset += (go_all && nb) ? (64u << n_log2c) : 0u;
        log2c = go ? n_log2c : log2c;
        size = go ? n_size : size;
        kidx = go ? n_kidx : kidx;
        nzeros = go ? tok : nzeros;
        k = go ? covered : k;
        covm1 = go ? n_covm1 : covm1;
        if (ASMT) {
          covm2 = go ? n_covm1 - 2 : covm2;
          cbase = go ? n_cbase + L.ctx : cbase;
          dst = go ? uint32_t(((g * 3 + c) * 65536 + n_coef_offset + covered) * sizeof(CoefT)) : dst;  // position `covered` itself
          addr_a = go ? n_addr_a + L.ctx : addr_a;
        } else {
        cbase = go ? n_cbase : cbase;
        dst = go ? uint32_t(((g * 3 + c) * 65536 + n_coef_offset + (covered & ~(kPerChunk - 1))) * sizeof(CoefT)) : dst;  // chunk of position `covered`
        acc_lo = go ? 0u : acc_lo;
        acc_hi = go ? 0u : acc_hi;
        addr_a = go ? n_addr_a : addr_a;
        }
        nnz_b = go ? n_nnz_b : nnz_b;
        ctxe = go ? n_ctxe : ctxe;
        mode = (go && tok != 0 && !bad) ? uint32_t(kRun) : mode;
        err |= (go && bad) ? kErrNzeros : 0u;  // abandon the section: the refill round takes the "section complete" path
        bi = go_all ? ((go && bad) ? b1 : n_bi) : bi;
        ci = go_all ? ((go && bad) ? 2u : n_ci) : ci;
        }
        if (B.prof) {
          t_trans += __builtin_readcyclecounter() - t2;
          n_service++;
          t_dma += n_go;  // (lanes served)
        }
      }
    }
    // ================================================================= every few rounds: ring refills, rare transitions
    if ((++round & B.refill_mask) != 0) continue;
    {
      unsigned long long t0 = 0;
      if (B.prof) t0 = __builtin_readcyclecounter();
      if (!__ballot(mode != kDone)) break;
      // (1) what the previous refill round requested has landed (the wait also covers the coefficient stores since: a
      // write round trip, once per refill round)
      if constexpr (ASMT) {
        // the words and block records requested by the previous refill round sit in registers (one 16-byte load per lane):
        // into this lane's rows of the rings
        if (__ballot(stg_n | btg_n)) LaneLoadWait(stg, btg);
        if (__ballot(stg_n != 0)) {
          if (stg_n) {
            const bool zeros = stg_n == 8;
            const uint32_t s0 = zeros ? 0u : stg.x, s1 = zeros ? 0u : stg.y, s2 = zeros ? 0u : stg.z, s3 = zeros ? 0u : stg.w;
            LdsU32* const r4 = ring + (ring_end & (kLanesRingWords - 1)) * 64;  // (ring_end is a multiple of 4: the rows do not wrap)
            r4[0] = s0;
            r4[64] = s1;
            r4[128] = s2;
            r4[192] = s3;
            if ((ring_end & (kLanesRingWords - 1)) == 0) {  // mirror rows
              ring[16 * 64] = s0;
              ring[17 * 64] = s1;
            }
            ring_end += 4;
            stg_n = 0;
          }
        }
        if (__ballot(btg_n != 0)) {
          if (btg_n) {
            LdsU32* const r4 = bring + (bring_end & (kLanesBlockRing - 1) & ~3u) * 64;
            r4[0] = btg.x;
            r4[64] = btg.y;
            r4[128] = btg.z;
            r4[192] = btg.w;
            bring_end = (bring_end & ~3u) + 4;  // (a section's first request starts at the aligned block at or below its first one)
            btg_n = 0;
          }
        }
      } else {
      LaneDmaWait();
      ring_end += pend_s;
      bring_end += pend_b;
      pend_s = pend_b = 0;
      }
      // (2) the rare transitions (section header, section complete) behind a wave-uniform branch
      {
        const bool next_block = ci == 2 && started;
        const bool rare = mode == kWait && (ring_end - (bitpos >> 5)) >= 3 && (!started || (next_block && bi + 1 >= b1));
        if (__ballot(rare)) {
          if (rare) {
            if (!started) {  // section header: histogram selector + initial rANS state
              started = true;
              uint32_t hb = 0;
              while ((1u << hb) < num_hist) hb++;
              const uint32_t w0 = ring[0], w1 = ring[64];
              const uint64_t win = ((uint64_t(w1) << 32) | w0) >> bitpos;  // bitpos < 8 here
              const uint32_t sel = hb ? uint32_t(win) & ((1u << hb) - 1) : 0;
              if (sel != wg_sel) err = kErrSelector;  // (the host packed the section by the selector it read: cannot differ)
              state = PREFIX ? 0x13u << 16 : uint32_t(win >> hb);  // (prefix codes carry no state: what the end check expects)
              bitpos += hb + (PREFIX ? 0 : 32);
            } else {  // section complete (or abandoned after an error): on to the next one of the unit
              if (state != (0x13u << 16)) err |= kErrFinalState;
              if (bitpos > sec_size * 8) err |= kErrOverread;
              P.sec_end_bits[sec0 + g] = bitpos;
              // every section's flag word is written (the host does not clear the array); the passes of a progressive frame
              // share a group's word, which the host clears before the launch
              if (P.num_passes > 1) atomicOr(P.errors + g, err);
              else P.errors[g] = (B.debug & 2) ? ntok : err;
              ntok = 0;
              g = take_section();
              if (g == 0xFFFFFFFFu) mode = kDone;
              else open_section();
            }
          }
          __builtin_amdgcn_s_waitcnt(0);  // (no load of this branch stays in flight: the hot loop has no vmcnt wait of its own)
        }
      }
      // (3) request this round's ring refills: groups of four rows of the [slot][lane] rings, one LDS-DMA instruction per
      // row and only for the lanes whose next words belong there. Words past the section are zeros
      // (dec_bit_reader.h:84-144): those lanes write them themselves.
      if constexpr (ASMT) {
        // (3) this round's requests: four stream words / four block records per lane that has four free rows, as ONE 16-byte
        // load each into registers (the LDS-DMA form below costs an instruction per ring row and row group whatever the
        // number of lanes that need it: ~130 issue slots per round against ~30). Words past the section are zeros
        // (dec_bit_reader.h:84-144).
        const uint32_t used = ring_end - (bitpos >> 5);
        const bool want_s = mode != kDone && kLanesRingWords - used >= 4;
        const bool want_b = mode != kDone && bring_end < b1 && (bi + 1 + kLanesBlockRing - bring_end) >= 4;
        const unsigned long long ms = __ballot(want_s && ring_end < nwords), mb = __ballot(want_b);
        if (ms) LaneLoad128(stg, ms, P.sections, sec_off + ring_end * 4);
        if (mb) LaneLoad128(btg, mb, P.block_recs, (bring_end & ~3u) * 4);
        stg_n = want_s ? (ring_end < nwords ? 4u : 8u) : 0u;
        btg_n = want_b ? 4u : 0u;
      } else {
        const uint32_t used = ring_end - (bitpos >> 5);  // words of the ring still holding unread bits
        const uint32_t groups_s = mode == kDone ? 0u : (kLanesRingWords - used) >> 2;  // whole groups of 4 free rows
        const uint32_t first_g = (ring_end >> 2) & 3;
        const uint32_t next_b = bi + 1;  // records of blocks next_b .. stay in the ring
        // block records: groups of 4, aligned to 4 blocks (the record array is padded: jxlhip_frame_upload)
        const uint32_t bfirst = (bring_end >> 2) & 1;
        uint32_t groups_b = 0;
        if (mode != kDone && bring_end < b1) groups_b = (next_b + kLanesBlockRing - bring_end) >> 2;  // free groups (0..2)
        if (__ballot(groups_s != 0)) {
#pragma unroll
          for (uint32_t gi = 0; gi < 4; gi++) {
            const uint32_t d = (gi - first_g) & 3;  // the lane's d-th requested group lands in row group gi
            const bool in = d < groups_s;
            const uint32_t w = ring_end + d * 4;
            const bool beyond = w >= nwords;
            dma_group4(in && !beyond, lds_abs + ring_row0 + gi * 1024, P.sections, sec_off + w * 4);
            if (gi == 0) {  // mirror rows
              if (in && !beyond) {
                LaneDmaDword(lds_abs + ring_row0 + 16 * 256, P.sections, sec_off + w * 4);
                LaneDmaDword(lds_abs + ring_row0 + 17 * 256, P.sections, sec_off + w * 4 + 4);
              }
            }
            if (__ballot(in && beyond)) {
              if (in && beyond) {
                for (uint32_t j = 0; j < 4; j++) ring[(gi * 4 + j) * 64] = 0;
                if (gi == 0) ring[16 * 64] = ring[17 * 64] = 0;
              }
            }
          }
          pend_s = groups_s * 4;
        }
        if (__ballot(groups_b != 0)) {
#pragma unroll
          for (uint32_t gi = 0; gi < 2; gi++) {
            const uint32_t d = (gi - bfirst) & 1;
            const bool in = d < groups_b;
            dma_group4(in, lds_abs + bring_row0 + gi * 1024, P.block_recs, ((bring_end & ~3u) + d * 4) * 4);
          }
          pend_b = groups_b * 4 - (groups_b ? (bring_end & 3) : 0);  // (a group's first request starts at an aligned block)
        }
      }
      if (B.prof) t_service += __builtin_readcyclecounter() - t0;
    }
  }
  if (B.prof && lane == 0) {
    unsigned long long* o = B.prof + size_t(blockIdx.x * WPG + wave) * 8;
    o[0] = __builtin_readcyclecounter() - t_begin;
    o[1] = t_service;
    o[2] = n_service;
    o[3] = n_trips;
    o[5] = t_wait;
    o[6] = t_dma;
    o[7] = t_trans;
    if (B.debug & 32) o[6] = n_calls;  // (measurement aid: calls of the trip loop instead of the lanes served)
    if (B.debug & 16) {  // (measurement aid: where the wave ran: HW_ID | XCC_ID << 32 instead of the lanes served)
      uint32_t hw, xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      o[6] = (unsigned long long)hw | ((unsigned long long)xcc << 32);
    }
  }
  if (B.prof) {  // lane-trips of the wave (how full its trips were)
    unsigned long long total = n_lane_trips;
    for (int off = 32; off; off >>= 1) total += __shfl_down(total, off);
    if (lane == 0) B.prof[size_t(blockIdx.x * WPG + wave) * 8 + 4] = total;
  }
}

}  // namespace jxlhip
#endif  // JXL_HIP_ENTROPY_LANES_H_
