// libjxl_amd — lane-parallel AC entropy decode for gfx950 (MI355X): the throughput form of the entropy stage.
//
// Replaces (like k_entropy_uni) lib/jxl/dec_group.cc:469-542,594-639 (DecodeACVarBlock / GetBlockFromBitstream),
// dec_ans.h:170-257 (rANS symbol + hybrid uint), ans_common.h:102-142 (alias lookup), ac_context.h:63-143.
//
// One AC section (256x256 group) is an inherently serial adaptive-context rANS stream, so the only parallelism is
// ACROSS sections. Here every LANE of a wave decodes its own section: the decoder state (rANS state, bit position,
// block / coefficient cursors) lives in VGPRs, the frame's entropy tables are shared in LDS by the workgroup, and one
// trip of the hot loop decodes one coefficient token for every runnable lane. A wave-per-section scalar decoder
// (k_entropy_uni) is bound by the scalar ALU that the four SIMDs of a CU share; the vector ALUs give up to 64 decoders
// per wave for the same issue slots. The host packs sections of similar compressed size into the same wave (longest
// first) and chooses how many lanes per wave are populated, so that small batches still spread over all SIMDs.
//
// Memory behaviour, all chosen to keep global-memory latency out of the serial chain:
//   * each lane's compressed stream is staged through a 16-word ring in LDS ([slot][lane] layout: conflict-free);
//     the hot loop extracts bits with one ds_read2 + v_alignbit at the lane's bit position, the ring is topped up
//     with 16-byte global loads in the service phase only;
//   * coefficients are written in SCAN order (position k of the coefficient order, zeros included) with
//     fire-and-forget stores of aligned 8-byte chunks (four int16 at a time, collected in a register pair: a lane's
//     stores are uncoalesced by nature, and the memory pipeline serialises such a wave-instruction per lane), plus the
//     number of valid entries per (block, channel) in `kend`; the transform kernels apply the order permutation when
//     they stage a block into LDS. No zero fill of the coefficient buffer, no order-table lookup and no dependent load
//     in the decode loop.
//
// Control flow is a per-lane state machine kept convergent on the hot part:
//   RUN   a coefficient token is pending                       -> the hot trip (one shared code sequence)
//   WAIT  the lane finished a (block, channel), or its ring runs low -> serviced (block transition incl. the block's
//         non-zero-count symbol, ring refill) together with the other waiting lanes once enough of them wait
//   DONE  section finished, lane idle, or a stream error was flagged
#ifndef JXL_HIP_ENTROPY_LANES_H_
#define JXL_HIP_ENTROPY_LANES_H_

#include "jxl_hip_kernels.h"

namespace jxlhip {

struct EntropyLaneBatch {
  const EntropyParams* params;  // one per frame of the batch (device memory)
  const uint32_t* wg_unit;      // per workgroup: its unit = the sections of one frame that use one histogram set
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
  const uint8_t* wave_log_ls;   // per wave: log2 of its populated-lane capacity (lanes beyond it are idle); the wave's LDS
                                // rows are strided by that many entries, so sparse waves take little LDS
  uint32_t debug;               // measurement aid: bit 0 = skip the coefficient stores (results are then invalid),
                                // bit 1 = report every section's coefficient-token count in its error word
  unsigned long long* prof;     // optional (may be NULL): per wave {cycles total, cycles in service, services, hot trips}
};

// LDS layout of one workgroup (byte offsets, every region 16-byte aligned); the host sizes the launch with it.
struct LanesLds {
  uint32_t alias, ctx, ctx2, cfg, poff, wave0, per_wave, total;
};
// Per-wave LDS, all [row][lane] with a row stride of `lanes` entries (conflict-free, and a wave that populates few lanes
// needs little LDS, which keeps room on the CU for the bandwidth-bound kernels running beside this one):
constexpr uint32_t kLanesNzRows = 96;               // nzeros line buffer [channel * 32 + column], u8
constexpr uint32_t kLanesRingWords = 16;            // stream ring, u32; + 2 mirror rows (a 3-word read at slot 15 needs no wrap)
constexpr uint32_t kLanesBlockRing = 8;             // packed block records, u32
constexpr int kLanesTrips = 4;                      // hot trips per control check
constexpr uint32_t kLanesPerLaneBytes = kLanesNzRows + (kLanesRingWords + 2) * 4 + kLanesBlockRing * 4;
// alias_lds = false: the alias tables stay in global memory (k_entropy_lanes<..., GALIAS = true>); prefix = true: prefix
// codes (no alias tables at all; the per-cluster table offsets get a 1 KB region).
__host__ __device__ inline LanesLds LanesLdsLayout(uint32_t num_hist, uint32_t nctx, uint32_t num_clusters, uint32_t log_alpha,
                                                   uint32_t waves, uint32_t lanes, bool alias_lds = true, bool prefix = false) {
  LanesLds l;
  l.alias = 0;
  l.ctx = alias_lds ? (num_clusters << log_alpha) * 8 : 0;
  l.ctx2 = l.ctx + ((num_hist * nctx + 16 + 15) & ~15u);
  l.cfg = l.ctx2 + 64 * 2;
  l.poff = l.cfg + 256 * 2;
  l.wave0 = l.poff + (prefix ? 256 * 4 : 0);
  l.per_wave = kLanesPerLaneBytes * lanes;
  l.total = l.wave0 + waves * l.per_wave;
  return l;
}

// One rANS symbol + hybrid-uint extra bits for the calling lane from histogram `cluster` (alias tables start at LDS
// offset 0, 8 << log_alpha bytes per cluster; l_cfg[cluster] = split_exp | msb << 4 | lsb << 8). `ring` points at the
// lane's column of the stream ring.
// FLAT = true: no branches (the renormalisation and the extra bits are computed for every lane and selected), so that a
// hot trip is one basic block whose independent instructions the scheduler can interleave with the serial chain.
// GALIAS = true: the alias entry comes from `galias` (global memory, same packed form) instead of LDS offset 0: one
// cached global round trip on the serial chain per token, in exchange for 16-64 KB less LDS per frame.
// PREFIX = true: a prefix code instead of rANS (dec_huffman.h:28-41 in the two-level table form the host builds:
// l_poff[cluster] = first entry | root bits << 24, see PrefixLookup); `state` is unused.
template <bool FLAT = false, bool GALIAS = false, bool PREFIX = false>
__device__ __forceinline__ uint32_t LaneSymbol(uint32_t cluster, uint32_t& state, uint32_t& bitpos, const uint32_t* ring, uint32_t LS,
                                               uint32_t log_ls, const uint8_t* lds, const uint16_t* l_cfg, uint32_t log_entry,
                                               const uint2* galias = nullptr, const uint32_t* l_poff = nullptr,
                                               const uint32_t* ptable = nullptr) {
  const uint32_t ctxe = l_cfg[cluster];
  if (PREFIX) {
    const uint32_t po = l_poff[cluster];
    const uint32_t s0 = ((bitpos >> 5) & (kLanesRingWords - 1)) << log_ls;
    typedef const volatile __attribute__((address_space(3))) uint32_t* LdsVolatile;
    const uint32_t w0 = *(LdsVolatile)(ring + s0);
    const uint32_t w1 = *(LdsVolatile)(ring + s0 + LS);
    const uint32_t w2 = *(LdsVolatile)(ring + s0 + 2 * LS);
    const uint32_t boff = bitpos & 31;
    const uint32_t win = __builtin_amdgcn_alignbit(w1, w0, boff);
    const uint32_t e = PrefixLookup(ptable + (po & 0xFFFFFFu), po >> 24, win);
    uint32_t tok = e >> 8;
    const uint32_t len = e & 0xFFu;
    bitpos += len;
    const uint32_t boff2 = boff + len;  // < 47
    const uint32_t se = ctxe & 15;
    const bool take = tok >= (1u << se);
    if (FLAT && !__builtin_amdgcn_ballot_w64(take)) return tok;
    if (take) {
      const uint32_t msb = (ctxe >> 4) & 15, lsb = (ctxe >> 8) & 15;
      const uint32_t nb = (se - (msb + lsb) + ((tok - (1u << se)) >> (msb + lsb))) & 31u;
      const uint32_t low = tok & ((1u << lsb) - 1), top = tok >> lsb;
      const bool up = boff2 >= 32;
      const uint32_t xb = __builtin_amdgcn_alignbit(up ? w2 : w1, up ? w1 : w0, boff2 & 31) & ((1u << nb) - 1);
      bitpos += nb;
      tok = (((((1u << msb) | (top & ((1u << msb) - 1))) << nb) | xb) << lsb) | low;
    }
    return tok;
  }
  // the bit window (96 bits: 16 renormalisation bits + up to 31 extra bits from any bit offset) is read unconditionally
  // and up front (volatile: not sunk into the branches), so that it shares one LDS round trip with the alias entry;
  // LS is a power of two
  const uint32_t s0 = ((bitpos >> 5) & (kLanesRingWords - 1)) << log_ls;
  typedef const volatile __attribute__((address_space(3))) uint32_t* LdsVolatile;
  const uint32_t w0 = *(LdsVolatile)(ring + s0);
  const uint32_t w1 = *(LdsVolatile)(ring + s0 + LS);
  const uint32_t w2 = *(LdsVolatile)(ring + s0 + 2 * LS);
  const uint32_t boff = bitpos & 31;
  const uint32_t res = state & 0xFFFu, slot = res >> log_entry, pos = res & ((1u << log_entry) - 1);
  const uint2 e = GALIAS ? galias[(cluster << (12 - log_entry)) + slot]
                         : *reinterpret_cast<const uint2*>(lds + (cluster << (15 - log_entry)) + slot * 8);  // 8 << log_alpha per cluster
  const bool gt = pos >= (e.x >> 24);
  const uint32_t x = gt ? e.y : e.x;
  uint32_t tok = gt ? (x >> 24) : slot;
  const uint32_t hi = state >> 12;
  state = (x & 0xFFFu) * hi + hi + ((x >> 12) & 0xFFFu) + pos;
  const uint32_t win = __builtin_amdgcn_alignbit(w1, w0, boff);
  const bool need = state < (1u << 16);
  if (FLAT) {
    const uint32_t sh = need ? 16u : 0u;
    state = (state << sh) | (need ? (win & 0xFFFFu) : 0u);
    const uint32_t boff2 = boff + sh;  // < 48
    bitpos += sh;
    const uint32_t se = ctxe & 15;
    const bool take = tok >= (1u << se);
    // Tokens with extra bits are rare (|coefficient| >= 8 at the usual split of 16): the wave is bound by instruction
    // issue (one instruction per four cycles for a lone wave: rocprofv3 SQ_ACTIVE_INST_VALU = SQ_INSTS_VALU), so the
    // ~25 instructions of the extra-bits path are skipped when no lane needs them (a wave-uniform branch).
    if (!__builtin_amdgcn_ballot_w64(take)) return tok;
    const uint32_t msb = (ctxe >> 4) & 15, lsb = (ctxe >> 8) & 15;
    const uint32_t nb = (se - (msb + lsb) + ((tok - (1u << se)) >> (msb + lsb))) & 31u;
    const uint32_t low = tok & ((1u << lsb) - 1), top = tok >> lsb;
    const bool up = boff2 >= 32;
    const uint32_t xb = __builtin_amdgcn_alignbit(up ? w2 : w1, up ? w1 : w0, boff2 & 31) & ((1u << nb) - 1);
    const uint32_t big = (((((1u << msb) | (top & ((1u << msb) - 1))) << nb) | xb) << lsb) | low;
    bitpos += take ? nb : 0u;
    return take ? big : tok;
  }
  state = need ? ((state << 16) | (win & 0xFFFFu)) : state;
  const uint32_t boff2 = boff + (need ? 16u : 0u);  // < 48
  bitpos += need ? 16u : 0u;
  const uint32_t se = ctxe & 15;
  if (tok >= (1u << se)) {
    const uint32_t msb = (ctxe >> 4) & 15, lsb = (ctxe >> 8) & 15;
    const uint32_t nb = (se - (msb + lsb) + ((tok - (1u << se)) >> (msb + lsb))) & 31u;
    const uint32_t low = tok & ((1u << lsb) - 1), top = tok >> lsb;
    const bool up = boff2 >= 32;
    const uint32_t xb = __builtin_amdgcn_alignbit(up ? w2 : w1, up ? w1 : w0, boff2 & 31) & ((1u << nb) - 1);
    bitpos += nb;
    tok = (((((1u << msb) | (top & ((1u << msb) - 1))) << nb) | xb) << lsb) | low;
  }
  return tok;
}

// AIDS = false compiles the measurement aids (B.prof, B.debug) out. GALIAS: see LaneSymbol.
template <typename CoefT, int WPG, bool AIDS, bool GALIAS = false, bool PREFIX = false>
__global__ __launch_bounds__(64 * WPG) void k_entropy_lanes(EntropyLaneBatch B_) {
  EntropyLaneBatch B = B_;
  if (!AIDS) {
    B.prof = nullptr;
    B.debug = 0;
  }
  extern __shared__ __align__(16) uint8_t lds_raw[];
  // This wave is a serial dependency chain that leaves most issue slots empty: it takes precedence over the
  // bandwidth-bound kernels that share its SIMD (they fill the gaps) whatever their age.
  if (B_.prio) __builtin_amdgcn_s_setprio(3);
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t unit = B.wg_unit[blockIdx.x];
  const uint4 unit_desc = B.units[unit];
  const uint32_t wg_desc = unit_desc.x;
  const uint32_t wg_sel = wg_desc >> 16;  // the histogram set this workgroup's sections use
  const EntropyParams& P = B.params[wg_desc & 0xFFFF];
  const uint32_t pass = unit_desc.w;  // the unit's sections belong to this pass (0 unless the frame is progressive)
  const PassDev& T = P.passes[pass];
  const uint32_t log_alpha = T.log_alpha, log_entry = 12 - log_alpha, nclusters = T.num_clusters;
  const uint32_t num_bctx = P.num_bctx, nctx = P.nctx, num_hist = P.num_hist;
  const uint8_t* wls = B.wave_log_ls + blockIdx.x * WPG;
  const uint32_t log_ls = wls[wave];
  const uint32_t LS = 1u << log_ls;
  uint32_t wave_off = 0;  // the per-wave regions are packed one after the other
  for (uint32_t w = 0; w < wave; w++) wave_off += kLanesPerLaneBytes << wls[w];
  const LanesLds L = LanesLdsLayout(1, nctx, nclusters, log_alpha, 0, 0, !GALIAS && !PREFIX, PREFIX);
  const uint2* const galias = T.alias_packed;
  uint32_t* const l_poff = reinterpret_cast<uint32_t*>(lds_raw + L.poff);  // (PREFIX only)
  const uint32_t* const ptable = T.prefix_table;
  uint2* l_alias = reinterpret_cast<uint2*>(lds_raw + L.alias);
  uint8_t* l_ctx = lds_raw + L.ctx;                                      // context -> histogram (cluster)
  uint16_t* l_cfg = reinterpret_cast<uint16_t*>(lds_raw + L.cfg);        // per cluster: split_exp | msb << 4 | lsb << 8
  uint16_t* l_nnz2 = reinterpret_cast<uint16_t*>(lds_raw + L.ctx2);  // [ceil(nzeros left / covered)] -> 2 * kCoeffNumNonzeroContext
  uint8_t* l_nz = lds_raw + L.wave0 + wave_off;                       // line buffer of the per-block nzeros prediction
  uint32_t* ring = reinterpret_cast<uint32_t*>(l_nz + kLanesNzRows * LS) + lane;                 // stream ring [slot][lane]
  uint32_t* bring = ring + (kLanesRingWords + 2) * LS;                                           // block records [slot][lane]

  // ---- stage the frame's tables (whole workgroup)
  {
    const uint32_t n_ctx = nctx + 16;  // the selected set's slice of the context map
    const uint8_t* ctx_slice = T.ctx_map + size_t(wg_sel) * nctx;
    for (uint32_t i = tid; i < n_ctx; i += 64 * WPG) {
      const uint32_t cl = ctx_slice[i];
      l_ctx[i] = uint8_t(cl < nclusters ? cl : nclusters - 1);
    }
    for (uint32_t i = tid; i < nclusters; i += 64 * WPG) {
      const uint32_t cfg = T.cfg[i];
      l_cfg[i] = uint16_t((cfg & 15) | (((cfg >> 8) & 15) << 4) | (((cfg >> 16) & 15) << 8));
    }
    // alias entry {cutoff u8, right u8, freq0 u16 | offsets1 u16, freq1 u16} ->
    //   x = (freq0 - 1) & 0xFFF | cutoff << 24                    taken when pos <  cutoff: symbol = slot, offset = pos
    //   y = (freq1 - 1) & 0xFFF | offsets1 << 12 | right << 24    taken when pos >= cutoff
    // (repacked by jxlhip_frame_upload: PassDev::alias_packed)
    if (PREFIX)
      for (uint32_t i = tid; i < nclusters; i += 64 * WPG) l_poff[i] = T.prefix_offset[i];
    const uint32_t n_alias = (GALIAS || PREFIX) ? 0u : nclusters << log_alpha;
    for (uint32_t i = tid; i < n_alias; i += 64 * WPG) l_alias[i] = galias[i];
    if (tid < 64) l_nnz2[tid] = uint16_t(uint32_t(c_coeff_nnz_ctx[tid]) * 2);
    uint32_t* z = reinterpret_cast<uint32_t*>(l_nz);
    for (uint32_t i = lane; i < kLanesNzRows * LS / 4; i += 64) z[i] = 0;
  }
  __syncthreads();

  // ---- per-lane section setup
  enum : uint32_t { kWait = 0, kRun = 1, kDone = 3 };
  auto take_section = [&]() -> uint32_t {
    const uint32_t idx = atomicAdd(B.queue + unit, 1u);
    return idx < unit_desc.z ? B.list[unit_desc.y + idx] : 0xFFFFFFFFu;
  };
  uint32_t g = lane < B.wave_lanes[blockIdx.x * WPG + wave] ? take_section() : 0xFFFFFFFFu;
  uint32_t mode = g == 0xFFFFFFFFu ? uint32_t(kDone) : uint32_t(kWait);
  uint32_t err = 0, ntok = 0;  // ntok: measurement aid (debug bit 1: the flag word reports the section's token count)
  uint32_t b1 = 0, bi = 0, ci = 2;
  const uint4* stream4 = reinterpret_cast<const uint4*>(P.sections);
  const uint4* const rec4 = reinterpret_cast<const uint4*>(P.block_recs);
  typedef uint32_t __attribute__((address_space(1)))* GU32W;  // (global, not generic: a flat store also ticks the LDS counter)
  const GU32W kend_out = (GU32W)(uintptr_t)(P.kend + size_t(pass) * P.kend_pass_stride);
  uint32_t nwords = 0, sec_size = 0, ring_end = 0, bring_end = 0, bitpos = 0, state = 0, ctx_base = 0;
  bool started = false;
  // block / channel cursor
  uint32_t info = 0, lbx = 0, lby = 0, coef_offset = 0, next_offset = 0;
  const uint32_t sec0 = pass * P.num_groups;  // the pass's first entry in the section tables
  auto open_section = [&]() {  // cursors of section g (the nzeros line buffer needs no reset: every entry a section
                               // reads was written by an earlier block of the same section)
    bi = P.gbb[g] - 1;         // the first transition advances to the group's first block
    b1 = P.gbb[g + 1];
    bring_end = P.gbb[g] & ~3u;
    stream4 = reinterpret_cast<const uint4*>(P.sections + P.sec_word[sec0 + g]);  // 16-byte aligned (jxlhip_frame_upload)
    sec_size = P.sec_size[sec0 + g];
    nwords = (sec_size + 3) / 4;
    bitpos = sec0 + g == 0 ? P.first_bit_offset : 0;
    ring_end = 0;
    next_offset = 0;
    ci = 2;
    err = 0;
    started = false;
  };
  if (mode == kWait) open_section();
  // coefficient cursor
  // addr_a / addr_b: LDS byte address of the context entry of the NEXT coefficient at frequency context 0, if the
  // current token turns out zero (same non-zero count, prev = 0) / non-zero (one fewer to come, prev = 1)
  // (addr_b is kept as cbase + 1 + nnz_b with nnz_b the raw table value: the add happens where addr_b is used, so the
  // table read issued at the end of a trip is only waited for after the next trip has issued its other LDS reads)
  uint32_t nzeros = 0, k = 0, size = 0, log2c = 0, covm1 = 0, cbase = 0, addr_a = 0, nnz_b = 0, ctxe = 0, dbase = 0, kidx = 0;
  uint32_t acc_lo = 0, acc_hi = 0;  // the coefficient chunk in progress
  typedef CoefT __attribute__((address_space(1)))* GCoef;
  typedef uint32_t U32x2 __attribute__((ext_vector_type(2)));
  typedef U32x2 __attribute__((address_space(1)))* G64W;
  const GCoef coeffs = (GCoef)(uintptr_t)(static_cast<CoefT*>(P.coeffs) + P.coef_pass_base + size_t(pass) * P.coef_pass_stride);
  const uint32_t shift = T.shift;

  unsigned long long t_begin = 0, t_service = 0, n_service = 0, n_trips = 0, t_hot0 = 0, t_hot1 = 0, t_land = 0;
  if (B.prof) t_begin = __builtin_readcyclecounter();
  for (;;) {
    // a group of kLanesTrips hot trips consumes at most kLanesTrips * 47 bits (16 renormalisation + 31 extra bits each),
    // i.e. bits of at most 2 * kLanesTrips ring words (words read beyond ring_end are stale but never consumed)
    const bool low = mode != kDone && (ring_end - (bitpos >> 5)) < 2 * kLanesTrips + 1;
    const uint64_t runnable = __ballot(mode == kRun && !low);
    const uint64_t waiting = __ballot((mode == kWait) || low);
    if (!(waiting | runnable)) break;
    if (waiting && (!runnable || (uint32_t(__popcll(waiting)) << B.wait_shift) >= uint32_t(__popcll(runnable)))) {
      // ================================================================= service phase
      unsigned long long t0 = 0;
      if (B.prof) t0 = __builtin_readcyclecounter();
      // (1) issue this phase's ring refills first: their HBM latency overlaps the transition work below, and no load
      // is left in flight when the hot loop resumes (a pending load would make the compiler drain vmcnt, i.e. wait for
      // the previous iteration's coefficient stores, at the top of every hot-loop iteration)
      bool want_s = mode != kDone && (ring_end - (bitpos >> 5)) <= kLanesRingWords - 4;
      bool want_b = mode != kDone && bring_end < b1 && bring_end + 4 - (bi + 1) <= kLanesBlockRing;
      uint4 pf_s = make_uint4(0, 0, 0, 0), pf_b = make_uint4(0, 0, 0, 0);
      if (want_s && ring_end < nwords) pf_s = stream4[ring_end >> 2];  // (past the section the ring is fed zeros)
      if (want_b) pf_b = rec4[bring_end >> 2];
      // (2) block / channel transitions of the waiting lanes, including the block's non-zero-count symbol. A lane whose
      // channel turns out empty (non-zero count 0: common for the chroma channels) is still waiting afterwards, so up to
      // three transitions (a whole block) are made per phase.
      for (int pass = 0; pass < 3; pass++) {
      const bool next_block = ci == 2 && started;  // the coming transition moves on to block bi + 1
      const bool go = mode == kWait && (ring_end - (bitpos >> 5)) >= 3 && (!next_block || bi + 1 < bring_end || bi + 1 >= b1);
      if (pass && uint32_t(__popcll(__ballot(go))) < B.extra_pass_min) break;
      if (go) {
        if (!started) {  // section header: histogram selector + initial rANS state
          started = true;
          uint32_t hb = 0;
          while ((1u << hb) < num_hist) hb++;
          const uint32_t w0 = ring[0], w1 = ring[LS];
          const uint64_t win = ((uint64_t(w1) << 32) | w0) >> bitpos;  // bitpos < 8 here
          uint32_t sel = hb ? uint32_t(win) & ((1u << hb) - 1) : 0;
          if (sel != wg_sel) err = kErrSelector;  // (the host packed the section by the selector it read: cannot differ)
          ctx_base = 0;
          state = PREFIX ? 0x13u << 16 : uint32_t(win >> hb);  // (prefix codes carry no state: what the end check expects)
          bitpos += hb + (PREFIX ? 0 : 32);
        } else {
          ci++;
          if (ci >= 3) {
            ci = 0;
            bi++;
          }
          if (bi >= b1) {  // section complete (or abandoned after an error): on to the next one of the unit
            if (state != (0x13u << 16)) err |= kErrFinalState;
            if (bitpos > sec_size * 8) err |= kErrOverread;
            P.sec_end_bits[sec0 + g] = bitpos;
            // every section's flag word is written (the host does not clear the array); the passes of a progressive frame
            // share a group's word, which the host clears before the launch
            if (P.num_passes > 1) atomicOr(P.errors + g, err);
            else P.errors[g] = (B.debug & 2) ? ntok : err;
            ntok = 0;
            g = take_section();
            if (g == 0xFFFFFFFFu) {
              mode = kDone;
            } else {
              open_section();
              want_s = want_b = false;  // the loads issued above belong to the finished section
            }
          } else {
            if (ci == 0) {  // packed record (jxlhip_frame_upload): column | not first row << 5 | log2 covered_x << 6 |
                            // log2 covered_y << 9 | block context of X, Y, B << 12, 16, 20
              info = bring[(bi & (kLanesBlockRing - 1)) * LS];
              lbx = info & 31;
              lby = (info >> 5) & 1;
              log2c = ((info >> 6) & 7) + ((info >> 9) & 7);
              coef_offset = next_offset;  // blocks of a group are contiguous in its coefficient planes
              next_offset += 64u << log2c;
            }
            const uint32_t c = ci == 0 ? 1u : (ci == 1 ? 0u : 2u);
            const uint32_t cx = 1u << ((info >> 6) & 7);
            const uint32_t bctx = (info >> (12 + 4 * c)) & 15;
            uint8_t* line = l_nz + (c * 32) * LS + lane;
            uint32_t pred;
            if (lbx == 0) pred = lby ? line[0] : 32;
            else if (lby == 0) pred = line[(lbx - 1) * LS];
            else pred = (uint32_t(line[lbx * LS]) + line[(lbx - 1) * LS] + 1) >> 1;
            uint32_t nzb = pred >= 64 ? 64 : pred;
            nzb = nzb < 8 ? nzb : 4 + nzb / 2;
            const uint32_t tok = LaneSymbol<false, GALIAS, PREFIX>(l_ctx[ctx_base + nzb * num_bctx + bctx], state, bitpos, ring, LS, log_ls, lds_raw, l_cfg, log_entry, galias, l_poff, ptable);
            const uint32_t covered = 1u << log2c;
            size = covered * 64;
            kidx = bi * 3 + c;
            if (tok > size - covered) {
              err |= kErrNzeros;  // abandon the section: the next transition takes the "section complete" path
              bi = b1;
              ci = 2;
            } else {
              const uint8_t nzv = uint8_t((tok + covered - 1) >> log2c);
              for (uint32_t i = 0; i < cx; i++) line[(lbx + i) * LS] = nzv;
              if (tok == 0) {
                kend_out[kidx] = 0;  // stays waiting: next channel / block
              } else {
                nzeros = tok;
                k = covered;
                covm1 = covered - 1;
                cbase = L.ctx + ctx_base + num_bctx * 37 + 458 * bctx;
                dbase = (g * 3 + c) * 65536 + coef_offset;  // (a multiple of 64 entries: chunks are 8-byte aligned)
                acc_lo = acc_hi = 0;
                const uint32_t prev = nzeros > size / 16 ? 0 : 1;
                addr_a = cbase + l_nnz2[((nzeros + covm1) >> log2c) & 63];
                nnz_b = l_nnz2[((nzeros - 1 + covm1) >> log2c) & 63];
                ctxe = lds_raw[addr_a + prev];  // frequency context of k = covered is 0
                mode = kRun;
              }
            }
          }
        }
      }
      }
      // (3) land the refills in the LDS rings
      unsigned long long t3 = 0;
      if (B.prof) t3 = __builtin_readcyclecounter();
      if (want_s) {
        const uint32_t s = ring_end & (kLanesRingWords - 1);
        ring[(s + 0) * LS] = ring_end + 0 < nwords ? pf_s.x : 0;  // reads past the section are zeros
        ring[(s + 1) * LS] = ring_end + 1 < nwords ? pf_s.y : 0;
        ring[(s + 2) * LS] = ring_end + 2 < nwords ? pf_s.z : 0;
        ring[(s + 3) * LS] = ring_end + 3 < nwords ? pf_s.w : 0;
        if (s == 0) {  // mirror rows
          ring[kLanesRingWords * LS] = ring_end < nwords ? pf_s.x : 0;
          ring[(kLanesRingWords + 1) * LS] = ring_end + 1 < nwords ? pf_s.y : 0;
        }
        ring_end += 4;
      }
      if (want_b) {
        const uint32_t s = bring_end & (kLanesBlockRing - 1);
        bring[(s + 0) * LS] = pf_b.x;
        bring[(s + 1) * LS] = pf_b.y;
        bring[(s + 2) * LS] = pf_b.z;
        bring[(s + 3) * LS] = pf_b.w;
        bring_end += 4;
      }
      if (B.prof) {
        const unsigned long long now = __builtin_readcyclecounter();
        t_service += now - t0;
        t_land += now - t3;
        n_service++;
      }
      continue;
    }
    // =================================================================== hot trips: one coefficient token per lane each
    n_trips += kLanesTrips;
    unsigned long long th = 0;
    if (B.prof) th = __builtin_readcyclecounter();
#pragma unroll
    for (int rep = 0; rep < kLanesTrips; rep++) {
      if (B.prof && rep == 1) {
        const unsigned long long now = __builtin_readcyclecounter();
        t_hot0 += now - th;
        th = now;
      }
      if (mode == kRun && !low) {
        // context entries of coefficient k + 1 for both outcomes of this one (off the serial chain);
        // kCoeffFreqContext(b) for b = (k + 1) / covered in 1..63 is min(b - 1, 7 + b / 2, 15 + b / 4)
        const uint32_t kn = k + 1;
        const uint32_t b = kn >> log2c;
        const uint32_t f2 = min(min(b - 1, 7 + (b >> 1)), 15 + (b >> 2)) << 1;
        const uint32_t e_zero = lds_raw[addr_a + f2];
        const uint32_t addr_b = cbase + 1 + nnz_b;
        const uint32_t e_nonzero = lds_raw[addr_b + f2];
        // nnz table entry for the trip after next if this token is non-zero (if it is zero, nnz_b stays): read here, with
        // everything else, so that no LDS read is waited for at the end of the trip
        const uint32_t nnz_c = l_nnz2[((nzeros - 2 + covm1) >> log2c) & 63];
        const uint32_t tok = LaneSymbol<true, GALIAS, PREFIX>(ctxe, state, bitpos, ring, LS, log_ls, lds_raw, l_cfg, log_entry, galias, l_poff, ptable);
        const uint32_t sgn = uint32_t(-int32_t(tok & 1));  // odd token: negative
        const int32_t coeff = int32_t(((tok >> 1) ^ sgn) << shift);
        // Coefficients leave in aligned 8-byte chunks (4 x int16 / 2 x int32), shifted into a register pair from the top:
        // a lane's stores go to 64 different cache lines, and the memory pipeline takes such a wave-instruction one lane
        // at a time (~8 cycles each: scripts/ubench_trip.hip), so one 2-byte store per token caps the whole chip at
        // ~67 G tokens/s. The chunk of scan positions [4j, 4j + 4) is stored when position 4j + 3 has been decoded or the
        // (block, channel) ends; entries outside [covered, kend) of a chunk are unspecified (the transforms never read them).
        if (sizeof(CoefT) == 2) {
          acc_lo = __builtin_amdgcn_alignbit(acc_hi, acc_lo, 16);
          acc_hi = (acc_hi >> 16) | (uint32_t(coeff) << 16);
        } else {
          acc_lo = acc_hi;
          acc_hi = uint32_t(coeff);
        }
        if (B.debug & 2) ntok++;
        k = kn;
        const bool nz = tok != 0;
        nzeros -= nz ? 1u : 0u;
        ctxe = nz ? e_nonzero : e_zero;
        addr_a = nz ? addr_b - 1 : addr_a;
        nnz_b = nz ? nnz_c : nnz_b;
        constexpr uint32_t kPerChunk = 8 / sizeof(CoefT);
        const uint32_t part = k & (kPerChunk - 1);  // entries of the chunk in progress
        if (part == 0 || nzeros == 0) {
          uint32_t lo = acc_lo, hi = acc_hi;
          if (part) {  // a partial last chunk: its entries sit at the top of the pair
            const uint64_t v = ((uint64_t(hi) << 32) | lo) >> ((kPerChunk - part) * (64 / kPerChunk));
            lo = uint32_t(v);
            hi = uint32_t(v >> 32);
          }
          if (!(B.debug & 1)) *(G64W)(coeffs + (dbase + ((k - 1) & ~(kPerChunk - 1)))) = U32x2{lo, hi};
        }
        if (nzeros == 0) {
          kend_out[kidx] = k;
          mode = kWait;
        } else if (k >= size) {
          err |= kErrNzeros;  // abandon the section
          bi = b1;
          ci = 2;
          mode = kWait;
        }
      }
    }
    if (B.prof) t_hot1 += __builtin_readcyclecounter() - th;
  }
  if (B.prof && lane == 0) {
    unsigned long long* o = B.prof + size_t(blockIdx.x * WPG + wave) * 8;
    o[6] = t_land;
    o[4] = t_hot0;
    o[5] = t_hot1;
    o[0] = __builtin_readcyclecounter() - t_begin;
    o[1] = t_service;
    o[2] = n_service;
    o[3] = n_trips;
  }
}

}  // namespace jxlhip
#endif  // JXL_HIP_ENTROPY_LANES_H_
