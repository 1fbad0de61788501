// libjxl_amd host front-end for Modular (lossless) frames: turns one frame into the flat plan the device decodes
// (include/jxl_amd_hip.h, JxlHipModFrameDesc; kernels in csrc/hip/jxl_hip_modular.h).
//
// Follows the frame walk of reference lib/jxl/dec_frame.cc:135-434 for frame_header.encoding == kModular and
// lib/jxl/dec_modular.cc:209-425: headers, TOC, the global MA tree + histograms, and for EVERY Modular stream (stream 0
// in the DC global section, one per DC group, one per AC group) its group header: use_global_tree, weighted-predictor
// header, transforms, an optional local tree + histograms (encoding.cc:554-684). What it does NOT do is touch a sample:
// each stream becomes a descriptor (where its sample data starts, which rectangles of which channel buffers it fills,
// which tree and code it uses), and the inverse transforms become a list of device operations on channel buffers.
#ifndef JXH_MODFRAME_H_
#define JXH_MODFRAME_H_

#include <string>
#include <vector>

#include "jxh_bits.h"
#include "jxh_entropy.h"
#include "jxh_headers.h"
#include "jxh_modular.h"
#include "jxh_splines.h"

namespace jxh {

struct ModPlanRect {
  uint32_t buffer, x0, y0, w, h, sig;
};
struct ModPlanStream {
  uint32_t section, bit_offset, stream_id, first_channel_index, tree, code, first_rect, num_rects;
  int32_t wp[11];
  uint32_t uses_wp, num_props, dist_multiplier, max_width, num_samples;
};
struct ModPlanOp {
  uint32_t kind;  // 0 RCT, 1 palette, 2 horizontal unsqueeze, 3 vertical unsqueeze
  uint32_t buf[6];
  uint32_t x0, y0, w, h, param, nb, bit_depth;
};

struct ModFramePlan {
  ImageHeader ih;
  FrameHeader fh;
  FrameDim dim;
  std::vector<uint64_t> section_offset;  // byte offset of every TOC section inside the buffer handed to the parser
  std::vector<uint32_t> section_size;
  std::vector<MTree> trees;              // [0] = the frame's global tree (possibly empty), then stream-local ones
  std::vector<EntropyCode> codes;
  std::vector<std::pair<uint32_t, uint32_t>> buffers;  // channel buffers: (w, h)
  std::vector<ModPlanRect> rects;
  std::vector<ModPlanStream> streams;
  std::vector<ModPlanOp> ops;            // local (per group) operations first, then the frame's inverse transforms
  uint32_t out_buffer[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
  uint32_t num_color = 3, has_alpha = 0, alpha_bits = 8;
  std::vector<uint32_t> extra_buffer;    // per extra channel: the buffer that holds it after the inverse transforms
  size_t frame_end = 0;
  bool xyb = false;  // the colour channels are XYB integers (Y, X, B - Y): converted by the colour stage on the device
  float dc_quant[3] = {1.0f / 4096, 1.0f / 512, 1.0f / 256};
  Splines splines;  // frame flag kSplines: drawn over the colour channels before the sample conversion
  bool has_splines = false;
  // frame flag kPatches (dec_frame.cc:271-285): drawn over the colour channels before the splines, like on a VarDCT frame
  Patches patches;
  bool has_patches = false;
  std::vector<uint32_t> patch_records, patch_row_start, patch_row_list;
  const float* patch_src[4] = {nullptr, nullptr, nullptr, nullptr};  // device planes of the reference slots (set_patch_sources)
  uint32_t patch_src_w[4] = {0, 0, 0, 0}, patch_src_h[4] = {0, 0, 0, 0};
};

class ModFrameParser {
 public:
  ModFrameParser(const uint8_t* data, size_t size) : data_(data), size_(size) {}

  void ParseFrame(size_t pos, const ImageHeader& ih, ModFramePlan* plan, size_t frame_index = 0, size_t nonvisible_index = 0) {
    (void)frame_index;
    (void)nonvisible_index;  // (Modular frames carry no noise: nothing here depends on the frame's position)
    ModFramePlan& P = *plan;
    P.ih = ih;
    BitReader br(data_ + pos, size_ - pos);
    ReadFrameHeader(br, ih, &P.fh);
    const FrameHeader& fh = P.fh;
    JXH_CHECK(fh.modular, "not a Modular frame");
    // (1 = kDCFrame: kept before the colour transform as a later frame's DC image; 2 = kReferenceOnly; 3 = kSkipProgressive)
    // (where the frame sits on the canvas and how it blends: the caller's business, as for FrameParser::ParseFrame)
    JXH_CHECK(!fh.ycbcr, "unsupported: YCbCr Modular frames");
    JXH_CHECK(!(ih.xyb_encoded && ih.gray), "unsupported: grey XYB Modular frames");
    JXH_CHECK(fh.upsampling == 1 && fh.num_passes == 1, "unsupported: upsampled / multi-pass Modular frames");
    for (uint32_t u : fh.ec_upsampling) JXH_CHECK(u == 1, "unsupported: upsampled extra channels");
    JXH_CHECK(!(fh.flags & (FrameHeader::kNoise | FrameHeader::kUseDcFrame)), "unsupported: noise / kUseDcFrame on Modular frames");
    // float and > 16-bit colour samples: the streams and transforms are 32-bit integers with 64-bit predictors whatever the
    // depth; the pixel writer widens them (dec_modular.cc:128-185,633-690). Extra channels of such types are refused.
    JXH_CHECK(!(ih.floating && ih.xyb_encoded), "float samples in an XYB image");
    for (const auto& e : ih.extra) JXH_CHECK(!e.floating && e.bits <= 16, "unsupported: float or > 16-bit extra channels");
    P.dim = MakeFrameDim(fh);
    const FrameDim& d = P.dim;
    const size_t entries = d.num_groups == 1 ? 1 : 2 + d.num_dc_groups + d.num_groups;
    Toc toc;
    ReadToc(br, entries, &toc);
    JXH_CHECK(!br.Overread(), "truncated frame header");
    const size_t base = pos + br.BitPos() / 8;
    JXH_CHECK(base + toc.total <= size_, "truncated frame");
    P.frame_end = base + toc.total;
    for (size_t i = 0; i < entries; i++) {
      P.section_offset.push_back(base + toc.offset[i]);
      P.section_size.push_back(toc.size[i]);
    }
    // ---- DC global: DC dequant (read, unused), the global tree + histograms, stream 0
    P.trees.assign(1, MTree());
    P.codes.assign(1, EntropyCode());
    BitReader g(data_ + P.section_offset[0], P.section_size[0]);
    if (fh.flags & FrameHeader::kPatches) {  // dec_frame.cc:271-285: the dictionary comes first in DC global
      JXH_CHECK(!ih.gray, "unsupported: patches on grey images");
      DecodePatches(g, d.xsize, d.ysize, ih.extra.size(), &P.patches);
      // (a Modular frame's alpha channel lives with its integer channels: blending through it is not wired up)
      JXH_CHECK(!P.patches.uses_alpha || ih.extra.empty(), "unsupported: patches that blend through alpha on Modular frames");
      BuildPatchRows(P.patches, d.ysize, &P.patch_records, &P.patch_row_start, &P.patch_row_list);
      P.has_patches = true;
    }
    if (fh.flags & FrameHeader::kSplines) {  // dec_frame.cc:289-308; a Modular frame has the default colour correlation (0, 1)
      JXH_CHECK(!ih.gray, "unsupported: splines on grey images");
      DecodeSplines(g, d.xsize * d.ysize, &P.splines);
      InitSplineDrawCache(&P.splines, d.xsize, d.ysize, 0.0f, 1.0f);
      P.has_splines = true;
    }
    // dec_modular.cc:583-631: an XYB Modular frame codes Y, X, B - Y in units of the DC quantisation steps read here
    P.xyb = ih.xyb_encoded;
    if (!g.ReadBool())
      for (int c = 0; c < 3; c++) {
        P.dc_quant[c] = ReadF16(g) * (1.0f / 128.0f);
        JXH_CHECK(P.dc_quant[c] >= 1e-8f, "invalid DC quant");
      }
    bool have_global = false;
    if (g.ReadBool()) {
      const size_t nb = (ih.gray ? 1 : 3) + ih.extra.size();
      DecodeTree(g, &P.trees[0], std::min<size_t>(size_t(1) << 22, 1024 + d.xsize * d.ysize * nb / 16));
      DecodeHistograms(g, (P.trees[0].size() + 1) / 2, &P.codes[0]);
      have_global = true;
    }
    P.num_color = ih.gray ? 1 : 3;
    std::vector<VCh> full;
    for (size_t c = 0; c < P.num_color + ih.extra.size(); c++) full.push_back(NewChannel(&P, d.xsize, d.ysize, 0, 0));
    size_t nb_meta = 0;
    std::vector<MTransform> transforms;
    {
      // stream 0: the channels no larger than a group (and every meta channel); the transforms stay pending
      std::vector<size_t> members;
      ParseStreamHeader(g, &P, have_global, &full, &nb_meta, &transforms, /*local=*/false);
      for (size_t i = 0; i < full.size(); i++) {
        if (i >= nb_meta && (full[i].w > d.group_dim || full[i].h > d.group_dim)) break;
        members.push_back(i);
      }
      std::vector<ModPlanRect> rects;
      for (size_t i : members) rects.push_back({full[i].buffer, 0, 0, uint32_t(full[i].w), uint32_t(full[i].h), Sig(full[i])});
      FinishStream(g, &P, 0, 0, rects, have_global, uint32_t(members.empty() ? 0 : members[0]));
    }
    // ---- group streams
    size_t first_big = nb_meta;
    while (first_big < full.size() && full[first_big].w <= d.group_dim && full[first_big].h <= d.group_dim) first_big++;
    auto group_stream = [&](size_t section, size_t x0, size_t y0, size_t span, int min_shift, int max_shift, uint32_t stream_id) {
      std::vector<VCh> part;
      std::vector<ModPlanRect> rects;
      for (size_t c = first_big; c < full.size(); c++) {
        const VCh& fc = full[c];
        const int shift = std::min(fc.hshift, fc.vshift);
        if (shift < min_shift || shift > max_shift) continue;
        const size_t rx = x0 >> fc.hshift, ry = y0 >> fc.vshift;
        if (rx >= fc.w || ry >= fc.h) continue;
        const size_t rw = std::min(span >> fc.hshift, fc.w - rx), rh = std::min(span >> fc.vshift, fc.h - ry);
        if (!rw || !rh) continue;
        VCh v = fc;
        v.w = rw;
        v.h = rh;
        part.push_back(v);
        rects.push_back({fc.buffer, uint32_t(rx), uint32_t(ry), uint32_t(rw), uint32_t(rh), Sig(v)});
      }
      if (part.empty()) return;
      BitReader r(data_ + P.section_offset[section], P.section_size[section]);
      size_t part_meta = 0;
      std::vector<MTransform> local;
      ParseStreamHeader(r, &P, have_global, &part, &part_meta, &local, /*local=*/true);
      FinishStream(r, &P, uint32_t(section), stream_id, rects, have_global, 0);
      for (size_t i = local.size(); i-- > 0;) {  // local transforms: only RCT (no channel-list change), on the group's rectangles
        const MTransform& t = local[i];
        JXH_CHECK(t.id == 0, "unsupported on the GPU path: group-local palette / squeeze");
        JXH_CHECK(t.begin_c + 2 < rects.size(), "RCT: channel range");
        const ModPlanRect &a = rects[t.begin_c], &b = rects[t.begin_c + 1], &c = rects[t.begin_c + 2];
        JXH_CHECK(a.w == b.w && a.w == c.w && a.h == b.h && a.h == c.h && a.x0 == b.x0 && a.x0 == c.x0 && a.y0 == b.y0 && a.y0 == c.y0,
                  "RCT: channel rectangles differ");
        ModPlanOp op{};
        op.kind = 0;
        op.buf[0] = a.buffer;
        op.buf[1] = b.buffer;
        op.buf[2] = c.buffer;
        op.x0 = a.x0;
        op.y0 = a.y0;
        op.w = a.w;
        op.h = a.h;
        op.param = t.rct_type;
        P.ops.push_back(op);
      }
    };
    if (entries == 1) {
      // one section: every channel fits stream 0 (a frame this small has no channel larger than a group)
      JXH_CHECK(first_big == full.size(), "single-section frame with group-coded channels");
    } else {
      for (size_t gi = 0; gi < d.num_dc_groups; gi++) {
        const size_t gx = gi % d.xsize_dc_groups, gy = gi / d.xsize_dc_groups;
        group_stream(1 + gi, gx * d.dc_group_dim, gy * d.dc_group_dim, d.dc_group_dim, 3, 1000, uint32_t(1 + d.num_dc_groups + gi));
      }
      for (size_t gi = 0; gi < d.num_groups; gi++) {
        const size_t gx = gi % d.xsize_groups, gy = gi / d.xsize_groups;
        group_stream(2 + d.num_dc_groups + gi, gx * d.group_dim, gy * d.group_dim, d.group_dim, 0, 2,
                     uint32_t(1 + 3 * d.num_dc_groups + 17 + gi));
      }
    }
    // ---- the frame's inverse transforms as device operations on whole channel buffers
    for (size_t i = transforms.size(); i-- > 0;) InverseOps(&P, transforms[i], &full, &nb_meta);
    JXH_CHECK(full.size() == P.num_color + ih.extra.size() && nb_meta == 0, "channel count after the inverse transforms");
    for (size_t c = 0; c < P.num_color; c++) {
      JXH_CHECK(full[c].w == d.xsize && full[c].h == d.ysize, "colour channel size");
      P.out_buffer[c] = full[c].buffer;
    }
    for (size_t e = 0; e < ih.extra.size(); e++) {
      JXH_CHECK(full[P.num_color + e].w == d.xsize && full[P.num_color + e].h == d.ysize, "extra channel size");
      P.extra_buffer.push_back(full[P.num_color + e].buffer);
    }
    for (size_t e = 0; e < ih.extra.size(); e++)
      if (ih.extra[e].type == 0 && !P.has_alpha) {
        P.has_alpha = 1;
        P.alpha_bits = ih.extra[e].bits;
        P.out_buffer[P.num_color] = full[P.num_color + e].buffer;
      }
  }

 private:
  struct VCh {  // a channel of the (virtual) Modular image: which device buffer holds it
    uint32_t buffer;
    size_t w, h;
    int hshift, vshift;
  };
  static uint32_t Sig(const VCh& v) { return uint32_t((v.hshift + 2) * 64 + (v.vshift + 2)); }
  static VCh NewChannel(ModFramePlan* P, size_t w, size_t h, int hs, int vs) {
    JXH_CHECK(w < (size_t(1) << 30) && h < (size_t(1) << 30), "channel too large");
    P->buffers.push_back({uint32_t(w), uint32_t(h)});
    return VCh{uint32_t(P->buffers.size() - 1), w, h, hs, vs};
  }

  // GroupHeader (encoding.cc:554-600) + the shape effect of its transforms on the channel list + an optional local tree
  // and code. On return `br` stands at the first bit of the stream's sample data (unless the stream has no samples).
  void ParseStreamHeader(BitReader& br, ModFramePlan* P, bool have_global, std::vector<VCh>* ch, size_t* nb_meta, std::vector<MTransform>* transforms,
                         bool local) {
    (void)local;
    use_global_ = br.ReadBool();
    wp_ = WpHeader();
    if (!br.ReadBool()) {
      wp_.p1C = int32_t(br.Read(5));
      wp_.p2C = int32_t(br.Read(5));
      wp_.p3Ca = int32_t(br.Read(5));
      wp_.p3Cb = int32_t(br.Read(5));
      wp_.p3Cc = int32_t(br.Read(5));
      wp_.p3Cd = int32_t(br.Read(5));
      wp_.p3Ce = int32_t(br.Read(5));
      for (int i = 0; i < 4; i++) wp_.w[i] = uint32_t(br.Read(4));
    }
    const uint32_t nt = ReadU32(br, Val(0), Val(1), BitsOffset(4, 2), BitsOffset(8, 18));
    transforms->resize(nt);
    for (auto& t : *transforms) ReadTransform(br, &t);
    for (auto& t : *transforms) MetaShape(P, &t, ch, nb_meta);
    (void)have_global;
  }

  // Reads a local tree + code if the header asked for one, then records the stream.
  void FinishStream(BitReader& br, ModFramePlan* P, uint32_t section, uint32_t stream_id, const std::vector<ModPlanRect>& rects,
                    bool have_global, uint32_t first_channel_index) {
    size_t samples = 0, max_w = 0;
    std::vector<ModPlanRect> live;
    for (const ModPlanRect& r : rects) {
      samples += size_t(r.w) * r.h;
      max_w = std::max<size_t>(max_w, r.w);
    }
    if (samples == 0) return;  // (ModularDecode returns before any tree / histogram when nothing is to be decoded)
    uint32_t tree = 0, code = 0;
    if (!use_global_) {
      P->trees.emplace_back();
      P->codes.emplace_back();
      DecodeTree(br, &P->trees.back(), std::min<uint64_t>(1024 + samples, 1 << 20));
      DecodeHistograms(br, (P->trees.back().size() + 1) / 2, &P->codes.back());
      tree = uint32_t(P->trees.size() - 1);
      code = uint32_t(P->codes.size() - 1);
    } else {
      JXH_CHECK(have_global && !P->trees[0].empty(), "global tree requested but absent");
    }
    JXH_CHECK(!br.Overread(), "Modular stream header over-read");
    const MTree& T = P->trees[tree];
    int max_prop = 15;
    bool uses_wp = false;
    for (const TreeNode& n : T) {
      if (n.property >= 0) max_prop = std::max(max_prop, n.property);
      if (n.property == 15 || (n.property < 0 && n.predictor == 6)) uses_wp = true;
    }
    JXH_CHECK(max_prop < 32, "unsupported on the GPU path: MA tree referring to more than four previous channels");
    ModPlanStream s{};
    s.section = section;
    s.bit_offset = uint32_t(br.BitPos());
    s.stream_id = stream_id;
    s.first_channel_index = first_channel_index;
    s.tree = tree;
    s.code = code;
    s.first_rect = uint32_t(P->rects.size());
    s.num_rects = uint32_t(rects.size());
    const int32_t wp[11] = {wp_.p1C, wp_.p2C, wp_.p3Ca, wp_.p3Cb, wp_.p3Cc, wp_.p3Cd, wp_.p3Ce,
                            int32_t(wp_.w[0]), int32_t(wp_.w[1]), int32_t(wp_.w[2]), int32_t(wp_.w[3])};
    memcpy(s.wp, wp, sizeof(wp));
    s.uses_wp = uses_wp ? 1 : 0;
    s.num_props = max_prop < 16 ? 16 : uint32_t(16 + DivCeil(size_t(max_prop - 16 + 1), 4) * 4);
    s.dist_multiplier = uint32_t(max_w);
    s.max_width = uint32_t(max_w);
    s.num_samples = uint32_t(std::min<size_t>(samples, 0xFFFFFFFFu));
    P->rects.insert(P->rects.end(), rects.begin(), rects.end());
    P->streams.push_back(s);
  }

  // Effect of a transform on the channel list (MetaApply: transform.cc:102-131, palette.cc:26-60, squeeze.cc:456-517);
  // every channel whose shape changes gets a fresh buffer of its coded size.
  static void MetaShape(ModFramePlan* P, MTransform* t, std::vector<VCh>* ch, size_t* nb_meta) {
    auto check_equal = [&](size_t c1, size_t c2) {
      JXH_CHECK(c1 <= c2 && c2 < ch->size(), "transform: invalid channel range");
      JXH_CHECK(!(c1 < *nb_meta && c2 >= *nb_meta), "transform: range spans meta and non-meta channels");
      for (size_t c = c1 + 1; c <= c2; c++)
        JXH_CHECK((*ch)[c].w == (*ch)[c1].w && (*ch)[c].h == (*ch)[c1].h && (*ch)[c].hshift == (*ch)[c1].hshift &&
                       (*ch)[c].vshift == (*ch)[c1].vshift,
                   "transform: channels differ in size");
    };
    if (t->id == 0) {
      check_equal(t->begin_c, t->begin_c + 2);
    } else if (t->id == 1) {
      const size_t b = t->begin_c, e = t->begin_c + t->num_c - 1;
      check_equal(b, e);
      if (b >= *nb_meta) (*nb_meta)++;
      else {
        JXH_CHECK(e < *nb_meta, "palette: bad meta channel range");
        *nb_meta += 2 - t->num_c;
      }
      ch->erase(ch->begin() + b + 1, ch->begin() + e + 1);
      ch->insert(ch->begin(), NewChannel(P, t->nb_colors + t->nb_deltas, t->num_c, -1, -1));
    } else {
      if (t->steps.empty()) {
        std::vector<MChannel> shapes;
        for (const VCh& v : *ch) {
          MChannel m;
          m.w = v.w;
          m.h = v.h;
          m.hshift = v.hshift;
          m.vshift = v.vshift;
          shapes.push_back(m);
        }
        DefaultSqueeze(shapes, *nb_meta, &t->steps);
      }
      for (const SqueezeStep& q : t->steps) {
        const size_t n = ch->size(), b = q.begin_c, e = size_t(q.begin_c) + q.num_c;
        JXH_CHECK(q.num_c >= 1 && b < n && e <= n, "squeeze: invalid channel range");
        if (b < *nb_meta) {
          JXH_CHECK(e <= *nb_meta && q.in_place, "squeeze: bad meta channel range");
          *nb_meta += q.num_c;
        }
        size_t at = q.in_place ? e : n;
        for (size_t c = b; c < e; c++, at++) {
          VCh src = (*ch)[c];
          JXH_CHECK(src.w && src.h, "squeeze of an empty channel");
          JXH_CHECK(src.hshift <= 30 && src.vshift <= 30, "squeeze: too many steps");
          size_t rw = src.w, rh = src.h;
          if (q.horizontal) {
            src.w = (src.w + 1) / 2;
            rw -= src.w;
            if (src.hshift >= 0) src.hshift++;
          } else {
            src.h = (src.h + 1) / 2;
            rh -= src.h;
            if (src.vshift >= 0) src.vshift++;
          }
          (*ch)[c] = NewChannel(P, src.w, src.h, src.hshift, src.vshift);
          ch->insert(ch->begin() + at, NewChannel(P, rw, rh, src.hshift, src.vshift));
        }
      }
    }
  }

  // Inverse of a transform as device operations (rct.cc:97-147, palette.cc:62-202, squeeze.cc:128-385).
  static void InverseOps(ModFramePlan* P, const MTransform& t, std::vector<VCh>* ch, size_t* nb_meta) {
    if (t.id == 0) {
      const size_t m = t.begin_c;
      JXH_CHECK(m + 2 < ch->size(), "RCT: channel range");
      const VCh &a = (*ch)[m], &b = (*ch)[m + 1], &c = (*ch)[m + 2];
      JXH_CHECK(a.w == b.w && a.w == c.w && a.h == b.h && a.h == c.h, "RCT: channel sizes differ");
      if (t.rct_type == 0) return;
      ModPlanOp op{};
      op.kind = 0;
      op.buf[0] = a.buffer;
      op.buf[1] = b.buffer;
      op.buf[2] = c.buffer;
      op.w = uint32_t(a.w);
      op.h = uint32_t(a.h);
      op.param = t.rct_type;
      P->ops.push_back(op);
    } else if (t.id == 1) {
      JXH_CHECK(*nb_meta >= 1, "palette transform without palette");
      JXH_CHECK(t.nb_deltas == 0 && t.predictor == 0, "unsupported on the GPU path: palette with delta entries / predictor");
      const VCh pal = (*ch)[0];
      const size_t nb = pal.h, c0 = t.begin_c + 1;
      JXH_CHECK(c0 < ch->size() && nb >= 1 && nb <= 4, "palette: channel out of range");
      const VCh idx = (*ch)[c0];
      ModPlanOp op{};
      op.kind = 1;
      op.buf[0] = pal.buffer;
      op.buf[1] = idx.buffer;
      op.buf[2] = idx.buffer;  // the first output replaces the index channel
      for (size_t i = 1; i < nb; i++) {
        const VCh v = NewChannel(P, idx.w, idx.h, idx.hshift, idx.vshift);
        ch->insert(ch->begin() + c0 + i, v);
        op.buf[2 + i] = v.buffer;
      }
      op.w = uint32_t(idx.w);
      op.h = uint32_t(idx.h);
      op.nb = uint32_t(nb);
      op.param = uint32_t(pal.w);
      op.bit_depth = std::min<uint32_t>(P->ih.bits, 24);
      if (idx.w && idx.h) P->ops.push_back(op);
      if (c0 >= *nb_meta) (*nb_meta)--;
      else *nb_meta -= 2 - nb;
      ch->erase(ch->begin());
    } else {
      for (size_t k = t.steps.size(); k-- > 0;) {
        const SqueezeStep& q = t.steps[k];
        const size_t b = q.begin_c, e = size_t(q.begin_c) + q.num_c;
        JXH_CHECK(e <= ch->size(), "squeeze: invalid channel range");
        const size_t first_res = q.in_place ? e : ch->size() + b - e;
        JXH_CHECK(first_res >= e && first_res + q.num_c <= ch->size(), "squeeze: residual channels missing");
        if (b < *nb_meta) {
          JXH_CHECK(*nb_meta >= q.num_c, "squeeze: meta channel bookkeeping");
          *nb_meta -= q.num_c;
        }
        for (size_t c = b; c < e; c++) {
          const VCh a = (*ch)[c], r = (*ch)[first_res + (c - b)];
          if (q.horizontal) {
            JXH_CHECK(a.w == (a.w + r.w + 1) / 2 && a.h == r.h, "squeeze: channel sizes do not match");
            if (r.w == 0) {
              (*ch)[c].hshift--;
              continue;
            }
            const VCh out = NewChannel(P, a.w + r.w, a.h, a.hshift - 1, a.vshift);
            ModPlanOp op{};
            op.kind = 2;
            op.buf[0] = a.buffer;
            op.buf[1] = r.buffer;
            op.buf[2] = out.buffer;
            if (a.h) P->ops.push_back(op);
            (*ch)[c] = out;
          } else {
            JXH_CHECK(a.h == (a.h + r.h + 1) / 2 && a.w == r.w, "squeeze: channel sizes do not match");
            if (r.h == 0) {
              (*ch)[c].vshift--;
              continue;
            }
            const VCh out = NewChannel(P, a.w, a.h + r.h, a.hshift, a.vshift - 1);
            ModPlanOp op{};
            op.kind = 3;
            op.buf[0] = a.buffer;
            op.buf[1] = r.buffer;
            op.buf[2] = out.buffer;
            if (a.w) P->ops.push_back(op);
            (*ch)[c] = out;
          }
        }
        ch->erase(ch->begin() + first_res, ch->begin() + first_res + q.num_c);
      }
    }
  }

  const uint8_t* data_;
  size_t size_;
  bool use_global_ = true;
  WpHeader wp_;
};

}  // namespace jxh
#endif  // JXH_MODFRAME_H_
