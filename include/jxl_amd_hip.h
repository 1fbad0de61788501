/* libjxl_amd — thin extern "C" HIP layer for the JPEG XL VarDCT decode hot path on MI355X (gfx950).
 *
 * Plain C ABI: pointers and sizes only, no C++ or torch types. One JxlHipContext owns one HIP stream and the
 * device buffers of one frame; all calls on a context are asynchronous on its stream unless noted.
 *
 * This is the internal seam of the reference that the GPU path replaces:
 *   - jxlhip_run_entropy()      replaces the entropy half of DecodeGroup():
 *                               lib/jxl/dec_group.cc:469-542 (DecodeACVarBlock), :594-660 (GetBlockFromBitstream),
 *                               lib/jxl/dec_ans.h:170-257 (rANS + hybrid uint), lib/jxl/dec_bit_reader.h:84-144.
 *   - jxlhip_run_transform()    replaces the reconstruction half of DecodeGroupImpl():
 *                               lib/jxl/dec_group.cc:115-181 (dequant + chroma-from-luma), :433-450,
 *                               lib/jxl/dec_transforms-inl.h:456-818 (TransformToPixels, LowestFrequenciesFromDC),
 *                               lib/jxl/dct-inl.h:376-397.
 *   - jxlhip_run_filter_color() replaces RenderPipelineInput::Done() for the VarDCT stage list:
 *                               lib/jxl/render_pipeline/stage_gaborish.cc:56-100, stage_epf.cc:82-494,
 *                               stage_xyb.cc:80-92, stage_from_linear.cc:114-144, stage_write.cc:266-286,548-590,
 *                               low_memory_render_pipeline.cc:475-517 (edge mirroring).
 * The public drop-in boundary (JxlDecoder* / JxlParallelRunner) is declared in include/jxl/decode.h and
 * include/jxl/parallel_runner.h and is implemented on top of this layer.
 *
 * All functions return 0 on success, a negative hipError_t on a HIP failure, or a positive JXLHIP_ERR_* code.
 */
#ifndef JXL_AMD_HIP_H_
#define JXL_AMD_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JXLHIP_ERR_INVALID_ARGUMENT 1
#define JXLHIP_ERR_NO_FRAME 2
#define JXLHIP_ERR_STREAM 3 /* the entropy kernel flagged a corrupt AC section; see jxlhip_get_errors */
#define JXLHIP_ERR_UNSUPPORTED 4

typedef struct JxlHipContext JxlHipContext;

/* One varblock in the decode order of its 256x256 group. */
typedef struct JxlHipVarBlock {
  uint16_t bx, by;      /* absolute position of the top-left 8x8 block */
  uint8_t strategy;     /* AcStrategy raw value 0..26 */
  uint8_t quant_dc_ctx; /* DC-derived bucket of the block context map */
  uint16_t qf;          /* raw quant field 1..256 */
  uint32_t coef_offset; /* offset of the block's coefficients in its group's per-channel plane */
} JxlHipVarBlock;

/* Entropy-code tables of one pass (rANS only). */
typedef struct JxlHipPassDesc {
  uint32_t log_alpha;
  uint32_t num_clusters;
  const uint8_t* ctx_map;     /* num_histograms * num_ac_contexts + 16 */
  uint32_t ctx_map_size;
  const void* alias;          /* num_clusters << log_alpha entries of 8 bytes:
                                 u8 cutoff, u8 right_value, u16 freq0, u16 offsets1, u16 freq1 */
  const uint32_t* uint_cfg;   /* per cluster: split_exponent | msb_in_token << 8 | lsb_in_token << 16 */
  const uint16_t* orders;     /* coefficient orders of the used (bucket, channel) pairs */
  uint32_t orders_size;
  uint32_t order_offset[39];  /* [bucket * 3 + channel] -> first entry in `orders` */
  uint32_t shift;             /* left shift applied to this pass's coefficients */
  /* Prefix-coded streams (dec_huffman.h:28-41; libjxl's fastest efforts) instead of rANS, as two-level lookup tables
   * (like the reference's own decoder, dec_huffman.cc): per cluster prefix_offset[cluster] = first entry of its root
   * table | root index bits R << 24 (R = min(8, longest code); 0 = one symbol, no bits). Root entry, indexed by the next
   * R bits of the stream (first bit = bit 0): code of <= R bits: symbol << 8 | code length; longer codes: (second-level
   * table, relative to the root's first entry) << 8 | 0x80 | S; second-level entry, indexed by the S bits after the
   * first R: symbol << 8 | total code length. alias / log_alpha are unused then. */
  uint32_t use_prefix;
  const uint32_t* prefix_table;
  uint32_t prefix_table_size;
  const uint32_t* prefix_offset;
  /* LZ77 (dec_ans.h:288-353): tokens >= lz_min_symbol start a copy of lz_min_length + hybrid(lz_len_cfg) earlier values at
   * a distance read with cluster lz_dist_ctx. lz_len_cfg is packed like uint_cfg. */
  uint32_t lz77, lz_min_symbol, lz_min_length, lz_len_cfg, lz_dist_ctx;
} JxlHipPassDesc;

/* The draw cache of a frame's splines: segments of 8 floats {centre x, centre y, maximum distance, 1 / sigma,
 * sigma / 4 * intensity, colour X, Y, B} (splines.h:44-51); row y draws segments row_segments[row_start[y] ..
 * row_start[y + 1]) in that order. num_segments = 0: no splines. */
typedef struct JxlHipSplines {
  uint32_t num_segments, num_row_segments;
  const float* segments;
  const uint32_t* row_start; /* ysize + 1 entries */
  const uint32_t* row_segments;
} JxlHipSplines;

/* The patches of a frame (lib/jxl/dec_patch_dictionary.cc): 8 u32 per position {x, y, xsize, ysize, x0 and y0 in the
 * reference frame, reference slot, PatchBlendMode of the colour channels (0 none, 1 replace, 2 add, 3 multiply) | clamp
 * << 8}; row y applies positions row_list[row_start[y] .. row_start[y + 1]) in that order. The reference frames are XYB
 * planes on the device ([3][slot_h][slot_w], from jxlhip_canvas_xyb_source). num_positions = 0: no patches. */
typedef struct JxlHipPatches {
  uint32_t num_positions, num_row_entries;
  const uint32_t* records;
  const uint32_t* row_start; /* ysize + 1 entries */
  const uint32_t* row_list;
  const float* slot_planes[4];
  uint32_t slot_w[4], slot_h[4];
  /* Patches that blend through alpha or change the alpha channel (PatchBlendMode 4..7: blend above / below, alpha-weighted
   * add above / below, dec_patch_dictionary.h:32-58; blending.cc:40-190), for images whose one extra channel is alpha:
   * records[7] also carries the alpha channel's own mode << 16 | its clamp << 24; uses_alpha != 0: some record needs the
   * alpha planes: slot_alpha[slot] = the reference frame's alpha, slot_h x slot_w floats on the device (jxlhip_canvas_xyb_alpha),
   * and the frame's own alpha plane must have been set (jxlhip_set_alpha) before the filter stage runs; the stage then leaves
   * the blended alpha for the pixel writer, the canvas and jxlhip_download_alpha. premultiplied = ExtraChannelInfo::
   * alpha_associated. Not on upsampled frames. Without uses_alpha modes 4 / 5 replace and 6 / 7 add (an image without alpha:
   * blending.cc:154-168). */
  uint32_t uses_alpha, premultiplied;
  const float* slot_alpha[4];
} JxlHipPatches;

typedef struct JxlHipFrameDesc {
  uint32_t xsize, ysize;
  uint32_t xsize_blocks, ysize_blocks;
  uint32_t xsize_groups, num_groups;
  uint32_t num_passes;
  uint32_t coef_bits; /* 16 or 32: storage type of quantised coefficients */
  /* compressed AC sections; index pass * num_groups + group */
  const uint8_t* codestream;
  const uint64_t* section_offset;
  const uint32_t* section_size;
  uint32_t first_section_bit_offset; /* single-section frames only, else 0 */
  const JxlHipPassDesc* passes;
  /* varblocks */
  const JxlHipVarBlock* blocks;
  uint32_t num_blocks;
  const uint32_t* group_block_begin; /* num_groups + 1 */
  /* block context map */
  const uint8_t* block_ctx_lut; /* [c][order bucket][qf bucket][dc bucket] */
  uint32_t block_ctx_lut_size;
  uint32_t num_block_ctxs, num_dc_ctxs, num_qf_thresholds;
  uint32_t qf_thresholds[16];
  uint32_t num_histograms;
  /* dequantisation tables: for table kind k, channel c: dequant[dequant_offset[k] + c * dequant_size[k] + i] */
  const float* dequant;
  uint32_t dequant_floats;
  uint32_t dequant_offset[17], dequant_size[17];
  /* block-resolution planes */
  const float* dc;        /* 3 planes X, Y, B of xsize_blocks * ysize_blocks */
  const float* inv_sigma; /* xsize_blocks * ysize_blocks */
  const int8_t* ytox;     /* ceil(xsize_blocks / 8) * ceil(ysize_blocks / 8) */
  const int8_t* ytob;
  /* scalars */
  float inv_global_scale, x_dm, b_dm;
  float color_scale, base_corr_x, base_corr_b;
  float quant_biases[4];
  /* loop filter */
  int32_t gab;
  float gab_w[6]; /* per channel {w1, w2} */
  int32_t epf_iters;
  float epf_channel_scale[3];
  float epf_pass0_sigma_scale, epf_pass2_sigma_scale, epf_border_sad_mul;
  /* colour */
  float opsin_inv[9]; /* inverse opsin matrix, already scaled by 255 / intensity_target */
  float opsin_bias[3];
  /* the colour stage behind the filters: 0 = XYB -> linear sRGB -> sRGB transfer function, 1 = XYB -> linear sRGB;
   * frames of images that are not xyb_encoded (frame_header.h ColorTransform; dec_cache.cc:256-263): 2 = kYCbCr (full-range
   * BT.601, stage_ycbcr.cc:41-60: channels Cb, Y, Cr), 3 = kNone (the channels are the samples); no transfer function
   * follows either (the samples are in the image's own colour space) */
  int32_t linear_output;
  /* Band decode (one frame split over several GPUs by rows of 256x256 groups): produce only the pixel
   * rows of group rows [band_group_row_begin, band_group_row_end). The context then also decodes and transforms the
   * one group row above and below the band (the loop filters need up to 7 rows of their output), so no exchange
   * between devices is needed. 0, 0 = the whole frame. */
  uint32_t band_group_row_begin, band_group_row_end;
  /* Upsampling of the decoded frame by 2, 4 or 8 before the colour conversion (stage_upsampling.cc:49-282; 0 or 1 = none):
   * xsize / ysize above are the coded frame, out_xsize / out_ysize the image (in (factor * (frame - 1), factor * frame]),
   * upsampling_kernel the factor * factor 5x5 kernels, [factor * oy + ox][5 * (iy + 2) + ix + 2]. */
  uint32_t upsampling, out_xsize, out_ysize;
  const float* upsampling_kernel;
  /* Noise synthesis (frame flag kNoise; dec_noise.cc:43-164, render_pipeline/stage_noise.cc:64-310): has_noise != 0 adds
   * noise to the filtered X, Y, B before the colour conversion, strength from the pixel's intensity through the 8-point
   * LUT; the generators are seeded with the frame's visible / non-visible index and each 256x256 group's origin. Only
   * for frames that are not upsampled. */
  uint32_t has_noise;
  float noise_lut[8];
  uint32_t noise_frame_index[2];
  /* Splines (render_pipeline/stage_splines.cc; the draw cache of lib/jxl/splines.cc:661-768, built by the host): drawn
   * over the filtered planes before noise and colour conversion. Only for frames that are not upsampled. */
  JxlHipSplines splines;
  /* Patches: drawn over the filtered planes before the splines (dec_cache.cc:193-201). Not for upsampled frames. */
  JxlHipPatches patches;
  /* A frame whose bytes are not all there yet (JxlDecoderFlushImage; lib/jxl/dec_frame.cc:735-795 Flush): NULL, or one byte
   * per group, non-zero = the group's AC sections are missing. Such a group is not entropy-decoded (its section sizes must
   * be 0); its blocks are rendered from the DC image alone, as the reference draws them with zero passes. */
  const uint8_t* group_absent;
  /* The block-resolution stencils of the DC path on the device (lib/jxl/compressed_dc.cc:130-198 AdaptiveDCSmoothing,
   * lib/jxl/epf.cc:39-81 ComputeSigma), so that the host front-end does no per-block float work:
   *  dc_smoothing != 0: `dc` is the dequantised DC image BEFORE smoothing; the upload smooths it (dc_step = the DC
   *    quantisation step of X, Y, B, i.e. inv_global_scale / quant_dc * dc_quant[c]). 0: `dc` is used as it is (a caller
   *    that holds libjxl's own state passes its smoothed image, or the frame has kSkipAdaptiveDCSmoothing).
   *  inv_sigma == NULL and epf_iters > 0: 1 / sigma per block is computed by the upload from `sharpness` (one byte 0..7 per
   *    block, row-major like `dc`), the varblocks' qf, quant_scale (= global_scale / 65536), epf_quant_mul and epf_sharp_lut. */
  uint32_t dc_smoothing;
  float dc_step[3];
  /* DequantDC on the device (compressed_dc.cc:201-296): dc_quantised != NULL hands over the three coded integer planes
   * (X, Y, B; each xsize_blocks * ysize_blocks) instead of `dc` (which may then be NULL); the upload forms
   * Y = q_y * step_y, X = Y * dc_cfl_x + q_x * step_x, B = Y * dc_cfl_b + q_b * step_b with step = dc_step[c] / 2^e, e =
   * dc_extra_precision[DC group] (one byte per group of 256 x 256 blocks, row-major; NULL = 0). */
  const int32_t* dc_quantised;
  const uint8_t* dc_extra_precision;
  float dc_cfl_x, dc_cfl_b;
  /* kUseDcFrame (frame_header.h:348, passes_state.cc:62-77): the DC image is an earlier DC frame's output, resident on
   * this device as 3 planes of xsize_blocks * ysize_blocks floats (jxlhip_canvas_xyb_source of slot 4 + level); `dc` may
   * then be NULL and no smoothing runs. The planes must stay valid until the transform stage has run. */
  const float* dc_device;
  const uint8_t* sharpness;
  float quant_scale, epf_quant_mul;
  float epf_sharp_lut[8];
  /* Chroma subsampling of a YCbCr frame (frame_header.h:81-166 YCbCrChromaSubsampling; linear_output 2 only, else all 0):
   * channel c (Cb, Y, Cr) has (xsize_blocks >> chroma_hshift[c]) x (ysize_blocks >> chroma_vshift[c]) blocks, shifts 0 or 1.
   * Its block at (sx, sy) is coded with the varblock at (sx << hshift, sy << vshift) (dec_group.cc:568-578: the other
   * varblocks carry nothing for it), its DC samples (`dc_quantised` / `dc`) sit in the top-left part of its plane with the
   * plane's row stride, and DequantDC applies no chroma from luma (compressed_dc.cc:232-250). xsize_blocks / ysize_blocks
   * are then whole MCUs (multiples of 1 << the largest shift); only varblocks of one block; dc_smoothing must be 0
   * (dec_frame.cc:206-212). The decoded channel is upsampled in front of the loop filters
   * (render_pipeline/stage_chroma_upsampling.cc:29-111, dec_cache.cc:138-150). */
  uint8_t chroma_hshift[3], chroma_vshift[3];
} JxlHipFrameDesc;

int jxlhip_device_count(void);
int jxlhip_ctx_create(int device, JxlHipContext** ctx);
void jxlhip_ctx_destroy(JxlHipContext* ctx);

/* Copies every table and the AC sections of the frame to the device; everything the call keeps has been copied when it
 * returns (the host arrays may be released). Re-uses device buffers across frames when they fit. */
int jxlhip_frame_upload(JxlHipContext* ctx, const JxlHipFrameDesc* desc);

/* The three stages of the hot path. Inputs must be resident (jxlhip_frame_upload). */
int jxlhip_run_entropy(JxlHipContext* ctx);
int jxlhip_run_transform(JxlHipContext* ctx);
int jxlhip_run_filter_color(JxlHipContext* ctx);
/* Convenience: all three, in order. */
int jxlhip_run_all(JxlHipContext* ctx);
/* Entropy stage of `n` resident frames (contexts on one device) as ONE kernel launch on ctxs[0]'s stream, so that the
 * per-section serial decoders of all frames fill the machine together; the other contexts' streams are made to wait
 * for it. Stage time is then available from ctxs[0] (which == 0). Falls back to n separate launches when the frames
 * cannot share a launch. */
int jxlhip_run_entropy_batch(JxlHipContext* const* ctxs, size_t n);
/* Transform stage / filter+colour stage of `n` resident frames with one launch per kernel for the whole set (on
 * ctxs[0]'s stream; stage times from ctxs[0]). The frames must share the device and the coefficient storage type.
 * jxlhip_run_transform(ctx) and jxlhip_run_filter_color(ctx) are the n = 1 case. */
int jxlhip_run_transform_batch(JxlHipContext* const* ctxs, size_t n);
int jxlhip_run_filter_color_batch(JxlHipContext* const* ctxs, size_t n);

/* Blocks until the context's stream is idle. */
int jxlhip_sync(JxlHipContext* ctx);

/* Copies the interleaved RGB8 result (row stride in bytes) to host memory; synchronous. The result has the frame's
 * size, or out_xsize x out_ysize for an upsampled frame. */
int jxlhip_download_rgb8(JxlHipContext* ctx, uint8_t* dst, size_t stride);
/* Output pixel format of the filter + colour stage (reference render_pipeline/stage_write.cc:266-286,334-370,548-590;
 * call before jxlhip_frame_upload; default RGB8). data_type uses the JxlDataType values (0 = f32, 2 = u8, 3 = u16,
 * 5 = f16), num_channels 1..4 (2 and 4 carry alpha: jxlhip_set_alpha, else opaque), bits_per_sample the depth of the
 * unsigned types (0 = full width), big_endian swaps the bytes of multi-byte samples. RGB8 is written by every filter
 * kernel and RGB f32 by the row-streaming one (the d1.0 configuration); other formats cost one more pass over the planes. */
int jxlhip_set_output_format(JxlHipContext* ctx, uint32_t data_type, uint32_t num_channels, uint32_t bits_per_sample, int big_endian);
/* The pixels are written with the image's orientation undone (JxlOrientation / EXIF numbering 1..8; 1 = as coded), the
 * way the reference's write stage does it (render_pipeline/stage_write.cc:292-306,441-458,664-699): for 5..8 the rows
 * of the output are columns of the image (jxlhip_download_pixels then hands out xsize rows of ysize pixels). Call before
 * the upload. */
int jxlhip_set_output_orientation(JxlHipContext* ctx, uint32_t orientation);
/* 2- and 4-channel output: the colour samples are divided by max(alpha, 2^-26) as they are written, after the transfer
 * function (what JxlDecoderSetUnpremultiplyAlpha asks of an image whose alpha is associated:
 * render_pipeline/stage_write.cc:359-361,460-482, dec_frame.h:209-211). Call before the upload; 0 = off (the default). */
int jxlhip_set_output_unpremultiply(JxlHipContext* ctx, int on);
/* Alpha plane of the image (f32 in [0, 1], xsize * ysize, host memory; copied synchronously) for 2- and 4-channel output;
 * NULL = opaque again. */
int jxlhip_set_alpha(JxlHipContext* ctx, const float* alpha, uint32_t xsize, uint32_t ysize);
/* An extra channel that is coded smaller than the image (JxlHipFrameDesc-independent; dec_cache.cc:172-190, 203-212: the
 * upsampling stage, stage_upsampling.cc:49-282, on that one channel): `plane` = xsize * ysize floats in host memory, `kernels`
 * = factor * factor * 25 weights (factor 2, 4 or 8), the result out_xsize x out_ysize with out in (factor * (size - 1),
 * factor * size]. as_alpha != 0: the result becomes the context's alpha plane (as after jxlhip_set_alpha); host_out != NULL:
 * it is copied there (out_xsize * out_ysize floats). Synchronous. */
/* The context's alpha plane as the filter stage left it (the patches of the frame may have blended into it), out_xsize *
 * out_ysize floats to host memory; synchronous. */
int jxlhip_download_alpha(JxlHipContext* ctx, float* dst);
int jxlhip_upsample_plane(JxlHipContext* ctx, const float* plane, uint32_t xsize, uint32_t ysize, uint32_t factor, const float* kernels,
                          uint32_t out_xsize, uint32_t out_ysize, int as_alpha, float* host_out);
/* Copies the interleaved result in the format of jxlhip_set_output_format (row stride in bytes); synchronous. */
int jxlhip_download_pixels(JxlHipContext* ctx, void* dst, size_t stride);
/* Same for pixel rows [y_begin, y_end) only (dst receives y_end - y_begin rows): the rows a band context produced. */
int jxlhip_download_rgb8_rows(JxlHipContext* ctx, uint8_t* dst, size_t stride, uint32_t y_begin, uint32_t y_end);
/* Device pointer of the RGB8 result (xsize * 3 bytes per row, tightly packed). */
const uint8_t* jxlhip_rgb8_device_ptr(JxlHipContext* ctx);

/* Per-group error flags written by the entropy kernel: bit0 invalid nzeros, bit1 ANS final state,
 * bit2 section over-read, bit3 invalid histogram selector. Synchronous. `flags` has num_groups entries.
 * Returns JXLHIP_ERR_STREAM if any flag is set. */
int jxlhip_get_errors(JxlHipContext* ctx, uint32_t* flags, size_t n);

/* Where every AC section's coefficient stream ended: bits[pass * num_groups + group] = bit position from the section's
 * first byte (after the entropy stage; synchronous). Sections of frames with extra channels carry Modular data behind
 * the coefficients (reference dec_frame.cc:511-542), which the host front-end decodes from there. */
int jxlhip_get_section_end_bits(JxlHipContext* ctx, uint32_t* bits, size_t n);

/* Test/debug access to intermediates; synchronous copies to host.
 *   "coeffs"       quantised coefficients, int16 or int32 [num_groups][3][65536]
 *   "xyb_idct"     float [3][ysize_padded][xsize_padded] after the inverse transforms
 *   "xyb_filtered" float [3][ysize_padded][xsize_padded] after Gaborish/EPF (rows < ysize, columns < xsize valid;
 *                  only with option "keep_filtered")
 * Returns the number of bytes the buffer needs through *needed when dst is NULL. */
int jxlhip_download(JxlHipContext* ctx, const char* name, void* dst, size_t dst_size, size_t* needed);

/* Options (set before jxlhip_frame_upload): "keep_filtered" = 1 makes jxlhip_run_filter_color also store the filtered
 * XYB planes for jxlhip_download("xyb_filtered") (test aid; costs one extra plane set and 12 B/pixel of writes).
 * "filter_async" = 1 (on the FIRST context of a set; may be changed at any time): jxlhip_run_filter_color_batch launches
 * on that context's second stream, ordered after the work its first stream holds. The call sequence entropy_batch,
 * transform_batch, filter_color_batch, entropy_batch, ... on one frame set then overlaps every filter + colour launch
 * with the next entropy launch (which touches neither planes nor pixels): the LDS-hungry entropy and transform kernels
 * never share the GPU, the LDS-free filter fills the gaps of the latency-bound entropy kernel. The library keeps the
 * order: the next transform, upload, download or sync of any frame of the set waits for the filter launch.
 * "entropy_gate" = 1 on the FIRST context of a frame set, for callers that keep a set's batched entropy launch and other
 * sets' batched transform / filter launches in flight together: the entropy launch must get the machine first (its few,
 * LDS-heavy workgroups find no room behind a stream of small ones: 133 ms instead of 62), so the transform / filter
 * launches enqueued after it wait ON THE DEVICE (hipStreamWaitValue64) until its workgroups are resident. The wait has no
 * bound, hence off by default, and the library ignores the option when the runtime may run kernels one at a time
 * (AMD_SERIALIZE_KERNEL, HIP_LAUNCH_BLOCKING, GPU_MAX_HW_QUEUES=1, counter collection by rocprofv3).
 * "blocking_sync" = 1 (process-wide for the context's device): host threads that wait for the device sleep instead of
 * spinning; for servers that pipeline frames over more host threads than they have CPUs to spare. */
int jxlhip_set_option(JxlHipContext* ctx, const char* name, int value);

/* ---- Bands of one frame on several devices (multi-GPU split of very large frames; replaces nothing in the reference,
 * whose LowMemoryRenderPipeline keeps the rows between groups in one address space: low_memory_render_pipeline.cc:475-517).
 * jxlhip_set_option(ctx, "band_halo", 1) before the upload of a band: the context entropy-decodes and transforms ONLY the
 * band's own rows of groups; the rows of the neighbouring bands that its filters read (jxlhip_halo_rows: Gaborish 1,
 * EPF 2 / 3 / 6 more) are exchanged after the transform stage: the neighbour packs them (side 0 = its first rows,
 * side 1 = its last rows) into a dense [3][rows][padded xsize] f32 block of DEVICE memory, any device-to-device transport
 * (hipMemcpyPeerAsync, an RCCL send / recv) moves the block, and jxlhip_halo_unpack writes it beside the band (side 0 =
 * above it, side 1 = below it). Both calls run behind the context's transform stage and are complete when they return
 * (one host synchronisation per call); the filter stage follows them.
 * The _batch forms take every frame of a set (the contexts of one jxlhip_run_transform_batch call, in that order; block i
 * of `block_bytes` bytes at `device + i * block_bytes`) and never block the host: `transport_stream` is the hipStream_t the
 * caller's transport is enqueued on (for torch.distributed over RCCL: torch's current stream, which c10d orders its
 * communication stream against). pack: the copies follow the set's transform launch, and everything enqueued on
 * transport_stream after the call follows the copies. unpack: the copies follow everything enqueued on transport_stream
 * BEFORE the call (the receive), the set's filter launch follows the copies, and work enqueued on transport_stream after
 * the call (re-use of the blocks) follows them too. */
int jxlhip_halo_rows(JxlHipContext* ctx, uint32_t* rows);
int jxlhip_halo_pack(JxlHipContext* ctx, int side, void* dst_device, size_t dst_bytes);
int jxlhip_halo_unpack(JxlHipContext* ctx, int side, const void* src_device, size_t src_bytes);
int jxlhip_halo_pack_batch(JxlHipContext* const* ctxs, size_t n, int side, void* dst_device, size_t block_bytes, void* transport_stream);
int jxlhip_halo_unpack_batch(JxlHipContext* const* ctxs, size_t n, int side, const void* src_device, size_t block_bytes,
                             void* transport_stream);

/* Test entry: runs the colour stage alone (XYB -> linear RGB -> sRGB transfer function unless linear_output) on n
 * XYB triples, planar [3][n], with the opsin parameters of the frame the context last uploaded; interleaved f32 RGB out.
 * For the reference's closed-form colour tests (lib/jxl/opsin_image_test.cc) against the kernel itself. */
int jxlhip_debug_color(JxlHipContext* ctx, const float* xyb, size_t n, int linear_output, float* rgb);

/* Debug aid: with JXLHIP_GUARD=1 in the environment every device buffer of a context is allocated with a 4 KiB guard
 * band either side, filled with a pattern. Waits for the device, then *touched = 0 when every band is intact, else
 * (1-based buffer index << 2) | (1 = band before, 2 = band after) of the first buffer a kernel wrote next to. */
int jxlhip_check_guards(JxlHipContext* ctx, uint32_t* touched);

/* Memory sharing for pipelined frame sets (call before jxlhip_frame_upload): `ctx` keeps its inverse-transform output (the
 * 3 f32 XYB planes, 12 B/pixel: the largest buffer of a context) in `lender`'s plane buffer instead of allocating its
 * own; lender = NULL undoes it. The planes only live between a frame's transform and its filter + colour stage, so
 * frames whose downstream stages never run at the same time (e.g. the two frame sets of a software pipeline: one set is
 * in the entropy stage while the other is transformed and filtered) can share them. The library orders the launches: a
 * transform that overwrites shared planes waits for the filter launch that last read them, also across streams. The
 * lender must have uploaded a frame at least as large and must outlive the borrower's use. */
int jxlhip_share_planes(JxlHipContext* ctx, JxlHipContext* lender);

/* ---- Modular (lossless) frames: reference lib/jxl/dec_modular.cc:209-425,564-793, modular/encoding/encoding.cc:148-724,
 * modular/transform/{rct,palette,squeeze}.cc. The host front-end parses headers, trees and histograms and every stream's
 * group header; the sample decode, the inverse transforms and the conversion to output samples run on the device. ---- */
typedef struct JxlHipModTreeNode {  /* MA tree node (dec_ma.cc:107-159); property -1 = leaf */
  int32_t property, splitval;
  uint32_t lchild, rchild; /* leaves: lchild = context */
  uint32_t predictor;
  int32_t offset;
  uint32_t multiplier, pad;
} JxlHipModTreeNode;
typedef struct JxlHipModCode {  /* one entropy code: the global one or a stream's own */
  const uint8_t* ctx_map;
  uint32_t ctx_map_size, num_clusters;
  uint32_t use_prefix, log_alpha;
  const void* alias;             /* ANS: num_clusters << log_alpha entries of 8 bytes (as JxlHipPassDesc::alias) */
  const uint32_t* uint_cfg;      /* per cluster */
  const uint32_t* prefix_table;  /* prefix codes: as JxlHipPassDesc */
  uint32_t prefix_table_size;
  const uint32_t* prefix_offset;
  uint32_t lz77, lz_min_symbol, lz_min_length, lz_len_cfg, lz_dist_ctx;
} JxlHipModCode;
typedef struct JxlHipModRect {  /* a stream's channel: rectangle of channel buffer `buffer` */
  uint32_t buffer, x0, y0, w, h, sig;
} JxlHipModRect;
typedef struct JxlHipModStream {
  uint32_t section;      /* index into the frame's section list */
  uint32_t bit_offset;   /* sample data start, from the section's first byte */
  uint32_t stream_id, first_channel_index;
  uint32_t tree, code;   /* indices into trees / codes */
  uint32_t first_rect, num_rects;
  int32_t wp[11];
  uint32_t uses_wp, num_props, dist_multiplier, max_width, num_samples;
} JxlHipModStream;
typedef struct JxlHipModBuffer {  /* a channel buffer: w x h int32 samples (the device pool is laid out by the library) */
  uint32_t w, h;
} JxlHipModBuffer;
typedef struct JxlHipModOp {  /* one inverse-transform step, in execution order */
  uint32_t kind;          /* 0 = RCT on a rectangle, 1 = palette lookup, 2 = horizontal unsqueeze, 3 = vertical unsqueeze */
  uint32_t buf[6];        /* RCT: 3 channels; palette: palette, index, outputs (buf[2..]); unsqueeze: averages, residuals, output */
  uint32_t x0, y0, w, h;  /* RCT rectangle (whole channels: 0, 0, w, h) */
  uint32_t param, nb, bit_depth;  /* RCT type; palette channels, sample depth */
  uint32_t after_stream;  /* run after stream index + 1 streams' launch... 0xFFFFFFFF = after all streams (global transforms) */
} JxlHipModOp;
typedef struct JxlHipModFrameDesc {
  uint32_t xsize, ysize;
  const uint8_t* codestream;
  const uint64_t* section_offset;
  const uint32_t* section_size;
  uint32_t num_sections;
  const JxlHipModTreeNode* const* trees;
  const uint32_t* tree_size;
  uint32_t num_trees;
  const JxlHipModCode* codes;
  uint32_t num_codes;
  const JxlHipModBuffer* buffers;
  uint32_t num_buffers;
  const JxlHipModRect* rects;
  uint32_t num_rects;
  const JxlHipModStream* streams;
  uint32_t num_streams;
  const JxlHipModOp* ops;
  uint32_t num_ops;
  /* output: the buffers holding the final colour channels (1 or 3) and alpha (or 0xFFFFFFFF) */
  uint32_t out_buffer[4];
  /* bits: low byte = bits per colour sample (integers: 1..31), bits 8..15 = exponent bits when the samples are floats of
   * that width (0 = integers; image_metadata.cc BitDepth, dec_modular.cc:128-185); alpha_bits: integer alpha, 1..24 */
  uint32_t num_color, has_alpha, bits, alpha_bits;
  /* splines over the three colour channels (as floats, before the sample conversion); colour images only */
  JxlHipSplines splines;
  /* patches over the three colour channels, before the splines (dec_cache.cc:193-201), from the reference slots' planes; colour
   * images only; not the modes that blend through the frame's alpha channel (uses_alpha must be 0) */
  JxlHipPatches patches;
  /* XYB Modular frames (dec_modular.cc:583-631): the colour buffers hold Y, X, B - Y as integers in units of
   * xyb_factor[] = the DC quantisation steps of X, Y, B; the colour stage (opsin_inv scaled by 255 / intensity_target,
   * opsin_bias, linear_output as in JxlHipFrameDesc) makes the samples. */
  uint32_t xyb;
  float xyb_factor[3];
  float opsin_inv[9];
  float opsin_bias[3];
  int32_t linear_output;
} JxlHipModFrameDesc;
/* Copies a Modular frame's tables and sections to the device (the arrays may be released afterwards). */
int jxlhip_modular_upload(JxlHipContext* ctx, const JxlHipModFrameDesc* desc);
/* Decodes every stream (one lane each), undoes the transforms and writes the pixels in the format of
 * jxlhip_set_output_format; results through jxlhip_download_pixels / jxlhip_download_rgb8. Stage time: which = 0. */
int jxlhip_modular_run(JxlHipContext* ctx);
/* Same for `n` resident frames: ONE streams launch for all their streams. */
int jxlhip_modular_run_batch(JxlHipContext* const* ctxs, size_t n);
/* Per-stream status words (0 = ok) and end bit positions after a run; synchronous. Returns JXLHIP_ERR_STREAM if any is set. */
int jxlhip_modular_status(JxlHipContext* ctx, uint32_t* status, uint32_t* end_bits, size_t n);
/* Test access: a channel buffer's int32 samples (w * h) after the run; synchronous. */
int jxlhip_modular_download_buffer(JxlHipContext* ctx, uint32_t buffer, int32_t* dst, size_t n);

/* ---- Canvas: the frames of a multi-frame codestream composed on the device (lib/jxl/blending.cc, alpha.cc,
 * render_pipeline/stage_blending.cc; dec_cache.cc:268-290 for where it sits: after the colour conversion, in the output
 * colour space). A frame is decoded by a context into f32 x 4 pixels (jxlhip_set_output_format(ctx, 0, 4, 0, 0), no
 * orientation), blended into the canvas with the reference slot its header names, optionally kept in a slot, and the
 * canvas is what jxlhip_canvas_download converts to the caller's format. */
typedef struct JxlHipCanvas JxlHipCanvas;
int jxlhip_canvas_create(int device, uint32_t xsize, uint32_t ysize, uint32_t has_alpha, uint32_t alpha_premultiplied, JxlHipCanvas** canvas);
void jxlhip_canvas_destroy(JxlHipCanvas* canvas);
typedef struct JxlHipBlend {
  int32_t x0, y0;             /* FrameOrigin: where the frame's top-left sample sits on the canvas (may be outside) */
  uint32_t mode, alpha_mode;  /* BlendMode of colour / alpha: 0 replace, 1 add, 2 blend, 3 alpha-weighted add, 4 multiply */
  uint32_t source, alpha_source; /* reference slots 0..3 the backgrounds come from (never written: zeros) */
  uint32_t clamp, alpha_clamp;
  int32_t save_slot;          /* 0..3: the blended canvas is also kept in this slot; -1: not kept */
} JxlHipBlend;
/* Frames kept BEFORE the colour transform: copies the XYB planes `frame` holds after its last run (VarDCT: the filtered
 * planes; Modular: needs jxlhip_set_option(frame, "keep_xyb_planes", 1) before the upload) into an XYB slot of the canvas:
 * slots 0..3 = the reference slots (kReferenceOnly frames, the sources of patches), slots 4..7 = the DC frames of level
 * 1..4 (kDCFrame; the DC image of a later frame with kUseDcFrame: JxlHipFrameDesc::dc_device). jxlhip_canvas_xyb_source
 * hands the device planes out ([3][h][w]; NULL / 0 while the slot is empty); they stay valid until the slot is written
 * again or the canvas is destroyed. */
int jxlhip_canvas_save_xyb(JxlHipCanvas* canvas, JxlHipContext* frame, uint32_t slot);
int jxlhip_canvas_xyb_source(JxlHipCanvas* canvas, uint32_t slot, const float** planes, uint32_t* xsize, uint32_t* ysize);
/* The alpha plane kept with that slot's frame (ysize x xsize floats), or NULL when the frame had none. */
int jxlhip_canvas_xyb_alpha(JxlHipCanvas* canvas, uint32_t slot, const float** alpha);
/* Blends the pixels `frame` holds (its last run, f32 x 4) into the canvas. */
int jxlhip_canvas_blend(JxlHipCanvas* canvas, JxlHipContext* frame, const JxlHipBlend* blend);
/* The canvas in a sample format (as jxlhip_set_output_format) and orientation (as jxlhip_set_output_orientation), into
 * host memory; rows of `stride` bytes. Synchronous. */
int jxlhip_canvas_download(JxlHipCanvas* canvas, uint32_t data_type, uint32_t num_channels, uint32_t bits_per_sample, int big_endian,
                           uint32_t orientation, void* dst, size_t stride);
/* jxlhip_canvas_download of a canvas with alpha hands out un-premultiplied colour (as jxlhip_set_output_unpremultiply) in its
 * 2- and 4-channel formats; the canvas itself stays as blended. 0 = off (the default). */
int jxlhip_canvas_set_unpremultiply(JxlHipCanvas* canvas, int on);
/* The canvas alpha plane as floats (xsize * ysize, not oriented). Synchronous. */
int jxlhip_canvas_download_alpha(JxlHipCanvas* canvas, float* dst, size_t n);

/* Test entry: the blend kernel alone on `n` caller-supplied samples: bg planar [4][n] (R, G, B, alpha), fg interleaved
 * [n][4], out planar [4][n]; for the reference's own blending vectors (lib/jxl/alpha_test.cc) against the kernel itself. */
int jxlhip_debug_blend(int device, const float* bg, const float* fg, size_t n, const JxlHipBlend* blend, uint32_t has_alpha,
                       uint32_t alpha_premultiplied, float* out);

/* ---- Forward path (first slice): the pixel-domain half of a VarDCT encode on the device.
 * Replaces, behind lib/jxl/enc_frame.cc:1135-1166's per-group loop: SRGBToXYB (enc_xyb.cc:152-174), the Gaborish
 * sharpening (enc_gaborish.cc:21-70) and ComputeCoefficients (enc_group.cc:380-533: forward transform, DC from the
 * lowest frequencies, AC quantisation with chroma-from-luma from the dequantised Y). Transform selection and the quant
 * field are this library's own simple activity heuristics, not the reference's search (enc_ac_strategy.cc,
 * enc_adaptive_quantization.cc): the output is a valid model of a frame, not libjxl's choice of one. Entropy coding
 * and the headers stay on the host (csrc/enc). */
typedef struct {
  uint32_t xsize, ysize;
  float distance;         /* Butteraugli-style target: scales the thresholds of the transform selection */
  uint32_t gaborish;      /* 1: pre-sharpen with the approximate inverse of the decoder's Gaborish blur */
  uint32_t strategy_mode; /* 0: DCT8 only, 1: activity-driven DCT 8..64 incl. rectangles */
  uint32_t global_scale, quant_dc; /* Quantizer parameters of the frame (quantizer.h:82-114) */
  float quant_ac;         /* AC quant target the quant field is built around */
  /* 1: fit the chroma-from-luma factors of every 64x64 tile like the reference's fast path (enc_chroma_from_luma.cc:
   * 128-151 FindBestMultiplier, 204-352 ComputeTile) and quantise X / B against them; 0: factors 0 (X), 0 (B on top of
   * the base 1.0). Output: ytox / ytob, one int8 per tile (ceil(xb / 8) * ceil(yb / 8)), host memory, may be NULL. */
  uint32_t cfl_fit;
  int8_t* ytox;
  int8_t* ytob;
  /* dequantisation tables, as JxlHipFrameDesc: kind k, channel c at dequant[dequant_offset[k] + c * dequant_size[k]] */
  const float* dequant;
  uint32_t dequant_floats;
  uint32_t dequant_offset[17], dequant_size[17];
} JxlHipEncDesc;
/* rgb: interleaved sRGB8 in host memory, `stride` bytes per row. Outputs (host memory): acs[yb * xb] = (strategy << 1) |
 * first-block bit, qf[yb * xb] (quant field at first blocks), dc[3][yb * xb] quantised DC stored X, Y, B, and
 * coeffs[groups][3][65536] quantised AC, block-contiguous per 256x256 group in raster order of the first blocks
 * (xb = ceil(xsize / 8), groups = ceil(xsize / 256) * ceil(ysize / 256)); coeffs may be NULL (the coefficients then stay
 * on the device: a caller that takes the tokens from jxlhip_enc_tokens needs no copy of them). Synchronous. */
int jxlhip_enc_forward(JxlHipContext* ctx, const uint8_t* rgb, size_t stride, const JxlHipEncDesc* desc, uint8_t* acs, int32_t* qf,
                       int32_t* dc, int32_t* coeffs);
/* Tokenisation of the last jxlhip_enc_forward's coefficients on the device (lib/jxl/enc_entropy_coder.cc:153-255
 * TokenizeCoefficients; csrc/hip/jxl_hip_enc.h): the (context, value) pairs the host entropy coder codes, in bitstream order
 * per 256x256 group (block by block in raster order of the first blocks, channels Y, X, B: the non-zero count, then the
 * coefficients from the first scan position behind the lowest-frequency corner to the last non-zero one). Single pass.
 *   orders: the coefficient orders of the 13 order buckets (one per bucket: the same for the three channels), bucket b at
 *   orders[order_offset[b]]; ctx_map: the block context of (channel, order bucket) at [(c < 2 ? c ^ 1 : 2) * 13 + bucket]
 *   (a block context map without quant-field / DC thresholds, ac_context.h:101-143); num_ctxs its number of contexts;
 *   num_hist: histogram sets (group g uses set g % num_hist: its contexts start at (g % num_hist) * num_ctxs * 495).
 * jxlhip_enc_token_counts computes everything but the tokens and returns every group's token count in totals[groups];
 * jxlhip_enc_tokens then writes group g's tokens at tokens[2 * bases[g] ...] as {context, value} pairs of uint32 (the
 * caller's prefix sum of the totals; `capacity` = number of pairs `tokens` holds). Both synchronous. */
typedef struct {
  const uint16_t* orders;
  uint32_t orders_size;
  uint32_t order_offset[13];
  uint8_t ctx_map[39];
  uint32_t num_ctxs, num_hist;
} JxlHipEncTokDesc;
int jxlhip_enc_token_counts(JxlHipContext* ctx, const JxlHipEncTokDesc* desc, uint32_t* totals);
int jxlhip_enc_tokens(JxlHipContext* ctx, const uint32_t* bases, uint32_t* tokens, size_t capacity);
/* Measurement: runs the kernel sequence of the last jxlhip_enc_forward `times` more times on its input, which is still
 * resident on the device (no copies); jxlhip_enc_last_ms then gives the time of all `times` passes. Synchronous. */
int jxlhip_enc_forward_rerun(JxlHipContext* ctx, uint32_t times);
/* Kernel time of the last jxlhip_enc_forward (HIP events around the launches, copies excluded), milliseconds. */
int jxlhip_enc_last_ms(JxlHipContext* ctx, float* ms);
/* The forward-transform kernel alone (the last pass of the last call), milliseconds. */
int jxlhip_enc_last_transform_ms(JxlHipContext* ctx, float* ms);

/* Timing of the last run of each stage in milliseconds (HIP events on the context's stream);
 * which: 0 entropy, 1 transform, 2 filter+colour. Synchronous. */
int jxlhip_last_stage_ms(JxlHipContext* ctx, int which, float* ms);

const char* jxlhip_version(void);

#ifdef __cplusplus
}
#endif
#endif /* JXL_AMD_HIP_H_ */
