/* libjxl_amd — frame-level C ABI between the host front-end and the HIP layer. Used by the JxlDecoder
 * implementation, by bench.py and by the parity tests to time the device stages separately from host parsing.
 *
 * jxlamd_frame_parse() does what the reference does once per frame on the host before any group is decoded
 * (lib/jxl/dec_frame.cc:135-434: headers, TOC, DC global, DC groups through the JxlParallelRunner, AC global);
 * jxlamd_frame_upload() hands the resulting tables and the still-compressed AC sections to a JxlHipContext. */
#ifndef JXL_AMD_H_
#define JXL_AMD_H_
#include <jxl/parallel_runner.h>
#include <stddef.h>
#include <stdint.h>

#include "jxl_amd_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
typedef struct JxlAmdFrame JxlAmdFrame;

/* Parses the first frame of a codestream (bare or in a `jxlc` container). `data` must stay valid until the frame
 * has been uploaded. runner may be NULL (sequential). Returns 0 or a non-zero code; see jxlamd_last_error(). */
int jxlamd_frame_parse(const uint8_t* data, size_t size, JxlParallelRunner runner, void* runner_opaque, JxlAmdFrame** frame);
/* The frame that starts at byte `frame_pos` of the buffer (the jxlamd_frame_end() of the one before it; 0 = the first),
 * `frame_index` its position in the codestream. Frames after the first exist in animations whose frames each replace the
 * whole canvas (decode.cc:1346-1350; full size, BlendMode kReplace, a duration): anything layered, blended, cropped or
 * referenced is refused. */
int jxlamd_frame_parse_at(const uint8_t* data, size_t size, size_t frame_pos, size_t frame_index, JxlParallelRunner runner,
                          void* runner_opaque, JxlAmdFrame** frame);
/* The same from a PREFIX of the frame's bytes (what JxlDecoderFlushImage draws from; lib/jxl/dec_frame.cc:735-795 Flush,
 * decode.cc:2458-2475): succeeds once the frame header, the TOC, the DC image (DC global + DC groups) and the AC global
 * section are whole. A group is drawn from its leading passes whose sections are whole (dec_frame.cc:620-680), from the DC
 * image alone when it has none; *groups_present = the number of groups with at least one pass. Fails ("truncated frame") for frames coded as a single
 * section and for frames with extra channels. jxlamd_frame_is_partial() tells such a frame from a whole one. */
int jxlamd_frame_parse_partial_at(const uint8_t* data, size_t size, size_t frame_pos, size_t frame_index, JxlParallelRunner runner,
                                  void* runner_opaque, JxlAmdFrame** frame, uint32_t* groups_present);
/* How many passes, from the first, EVERY group of the frame would have with `have_bytes` of the buffer there (a whole
 * frame, or one parsed from a prefix: its table of contents is known): what FrameDecoder::NumCompletePasses reports
 * (dec_frame.h:186-200) and JXL_DEC_FRAME_PROGRESSION steps by. info (optional, 10 values): num_passes, num_downsample,
 * downsample[4], last_pass[4] of the frame header's Passes (frame_header.h:286-309). */
uint32_t jxlamd_frame_complete_passes(const JxlAmdFrame* frame, size_t have_bytes, uint32_t* info);
int jxlamd_frame_is_partial(const JxlAmdFrame* frame);
/* Byte offset just behind the frame, and its animation fields {duration in ticks, is_last, timecode}. */
size_t jxlamd_frame_end(const JxlAmdFrame* frame, uint32_t* duration_last_timecode);
/* Where a frame sits on the canvas and how it combines with the reference slots (frame_header.h: FrameOrigin,
 * BlendingInfo, save_as_reference; blending.cc): what a caller needs to compose frames on a JxlHipCanvas. */
typedef struct {
  int32_t x0, y0;
  uint32_t xsize, ysize;
  uint32_t custom_size, frame_type; /* frame_type: 0 regular, 1 DC frame, 2 reference only (kept in XYB for patches), 3 skip-progressive */
  uint32_t mode, alpha_mode, source, alpha_source, clamp, alpha_clamp;
  uint32_t duration, is_last, save_as_reference, save_before_color_transform;
  /* DC frames (frame_header.h:319,348,439): dc_level 1..4 for a kDCFrame (its output, kept before the colour transform, is
   * the DC image of level dc_level - 1); use_dc_frame != 0: this frame's DC image is the DC frame of level dc_level + 1. */
  uint32_t dc_level, use_dc_frame;
} JxlAmdFramePlacement;
void jxlamd_frame_placement(const JxlAmdFrame* frame, JxlAmdFramePlacement* placement);
/* The struct above has grown (dc_level, use_dc_frame) and may grow again; it carries no size field. Callers that are not
 * compiled against THIS header (FFI mirrors, older binaries) use the sized forms: at most `out_size` bytes are written,
 * the return value is sizeof(JxlAmdFramePlacement) as this library knows it (also: jxlamd_sizeof_frame_placement). */
size_t jxlamd_sizeof_frame_placement(void);
size_t jxlamd_frame_placement_sized(const JxlAmdFrame* frame, void* placement, size_t out_size);
/* The frame's position among the shown / invisible frames (seeds its noise: dec_frame.cc:160-168); before upload. */
void jxlamd_frame_set_indices(JxlAmdFrame* frame, uint32_t visible_index, uint32_t nonvisible_index);
/* The reference frames a frame's patches read (device XYB planes of the four slots, jxlhip_canvas_xyb_source); checks every
 * patch rectangle against them. Before upload; returns non-zero (see jxlamd_last_error) when a patch refers to an empty
 * slot or reaches outside its reference frame. */
int jxlamd_frame_set_patch_sources(JxlAmdFrame* frame, const float* const* planes, const uint32_t* xsize, const uint32_t* ysize);
/* ... and their alpha planes (jxlhip_canvas_xyb_alpha; NULL entries: none), for patches that blend through alpha or write the
 * alpha channel (PatchBlendMode 4..7, or a mode on the alpha channel: blending.cc:40-190). Fails when such a patch names a
 * reference frame without alpha, when the frame is upsampled, or when the image's extra channel is not alpha. Call after
 * jxlamd_frame_set_patch_sources, before the upload. */
int jxlamd_frame_set_patch_alpha_sources(JxlAmdFrame* frame, const float* const* alpha);
/* A frame with kUseDcFrame: the device planes of the DC frame it names ([3][ysize][xsize] floats, e.g. from
 * jxlhip_canvas_xyb_source(canvas, 4 + placement.dc_level, ...)); fails unless the size is the frame's size in blocks. */
int jxlamd_frame_set_dc_source(JxlAmdFrame* frame, const float* planes, uint32_t xsize, uint32_t ysize);
void jxlamd_frame_free(JxlAmdFrame* frame);
/* info[0..15]: xsize, ysize, xsize_blocks, ysize_blocks, num_groups, num_dc_groups, num_passes, used_acs mask,
 * epf_iters, gab, coefficient storage bits (16/32), total AC section bytes, then of pass 0: log2 alphabet size,
 * number of clustered histograms, context map bytes; [15] reserved (0). */
void jxlamd_frame_info(const JxlAmdFrame* frame, uint32_t* info);
/* Byte sizes of the frame's AC group sections, [pass * num_groups + group]; returns their number (sizes may be NULL / short). */
size_t jxlamd_frame_section_sizes(const JxlAmdFrame* frame, uint32_t* sizes, size_t n);
/* Size of the decoded image: the frame size, times the upsampling factor of an upsampled frame (cropped to the image). */
void jxlamd_frame_out_size(const JxlAmdFrame* frame, uint32_t* width_height);
int jxlamd_frame_upload(const JxlAmdFrame* frame, JxlHipContext* ctx);
/* Same, for a band of rows of 256x256 groups [group_row_begin, group_row_end) (see JxlHipFrameDesc); 0, 0 = whole frame. */
int jxlamd_frame_upload_band(const JxlAmdFrame* frame, JxlHipContext* ctx, uint32_t group_row_begin, uint32_t group_row_end);
/* Output transfer function of the frame's pixels: 0 = sRGB, 1 = linear (JxlDecoderSetOutputColorProfile); before upload. */
void jxlamd_frame_set_linear_output(JxlAmdFrame* frame, int linear);
/* Extra channels (alpha, ...) of the frame. jxlamd_frame_extra_pending() != 0: their data continues behind the
 * coefficients of the AC group sections; run the entropy stage, then jxlamd_frame_finish_extra() (which reads the section
 * end positions from the context). jxlamd_frame_extra_plane() returns channel `index` as xsize * ysize int32 samples
 * (the frame's size), or NULL while pending / when there is no such channel. */
int jxlamd_frame_extra_pending(const JxlAmdFrame* frame);
int jxlamd_frame_finish_extra(JxlAmdFrame* frame, JxlHipContext* ctx);
/* The same with the groups' Modular streams spread over the caller's runner (NULL: serial). */
int jxlamd_frame_finish_extra_mt(JxlAmdFrame* frame, JxlHipContext* ctx, JxlParallelRunner runner, void* runner_opaque);
const int32_t* jxlamd_frame_extra_plane(const JxlAmdFrame* frame, uint32_t index);
/* An extra channel carries an upsampling factor of its own (frame_header.cc:265-283; dec_modular.cc:262-271): whu[0], whu[1] =
 * the size jxlamd_frame_extra_plane() has, ceil(image / factor); whu[2] = the factor (1, 2, 4, 8). jxlhip_upsample_plane
 * makes the image-sized plane of a factor > 1. Returns 0, or 1 for a bad argument. */
int jxlamd_frame_extra_dims(const JxlAmdFrame* frame, uint32_t index, uint32_t* whu);
/* The factor * factor 5x5 upsampling kernels of the image (its coded weights or the default ones: image_metadata.cc:87-214,
 * stage_upsampling.cc:59-84), factor * factor * 25 floats in JxlHipFrameDesc::upsampling_kernel's layout. */
int jxlamd_upsampling_kernels(const JxlAmdFrame* frame, uint32_t factor, float* kernels);
/* ---- Modular (lossless) frames: host parse into the plan the device decodes (csrc/host/jxh_modframe.h) ---- */
typedef struct JxlAmdModFrame JxlAmdModFrame;
/* Parses the first frame of a codestream as a Modular frame: headers, TOC, global tree and histograms and every stream's
 * group header; no sample is decoded on the host. `data` must stay valid until the frame has been uploaded. */
int jxlamd_modframe_parse(const uint8_t* data, size_t size, JxlAmdModFrame** frame);
/* The reference frames a Modular frame's patches read: as jxlamd_frame_set_patch_sources (before the upload). */
int jxlamd_modframe_set_patch_sources(JxlAmdModFrame* frame, const float* const* planes, const uint32_t* xsize, const uint32_t* ysize);
/* As jxlamd_frame_parse_at / jxlamd_frame_end, for Modular frames. */
int jxlamd_modframe_parse_at(const uint8_t* data, size_t size, size_t frame_pos, size_t frame_index, JxlAmdModFrame** frame);
size_t jxlamd_modframe_end(const JxlAmdModFrame* frame, uint32_t* duration_last_timecode);
void jxlamd_modframe_placement(const JxlAmdModFrame* frame, JxlAmdFramePlacement* placement);
size_t jxlamd_modframe_placement_sized(const JxlAmdModFrame* frame, void* placement, size_t out_size);
void jxlamd_modframe_free(JxlAmdModFrame* frame);
/* info[0..9]: xsize, ysize, colour channels, has alpha, bits per sample, streams, channel buffers, transform operations,
 * extra channels, compressed bytes of all sections. */
void jxlamd_modframe_info(const JxlAmdModFrame* frame, uint32_t* info);
/* Hands the plan to a context (jxlhip_modular_upload); then jxlhip_modular_run + jxlhip_download_pixels. */
int jxlamd_modframe_upload(const JxlAmdModFrame* frame, JxlHipContext* ctx);
/* Channel buffer of extra channel `index` after the run (jxlhip_modular_download_buffer), or 0xFFFFFFFF. */
uint32_t jxlamd_modframe_extra_buffer(const JxlAmdModFrame* frame, uint32_t index);
/* Thread-local description of the last failure of a jxlamd_* call ("" if none). */
const char* jxlamd_last_error(void);
/* Decodes a coded ICC profile: the entropy-coded, predicted byte stream that follows the headers of an image whose
 * ImageMetadata.color_encoding has want_icc (lib/jxl/icc_codec.cc:128-428). Returns 0 on success; *profile_size = bytes
 * of the profile, *coded_bits = exact length of the coded form; the profile is copied when out_size is large enough. */
int jxlamd_icc_decode(const uint8_t* coded, size_t size, uint8_t* out, size_t out_size, size_t* profile_size, size_t* coded_bits);

#ifdef __cplusplus
}
#endif
#endif /* JXL_AMD_H_ */
