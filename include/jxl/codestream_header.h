/* libjxl_amd: image/frame info structs (layout as reference lib/include/jxl/codestream_header.h:28-430). */
#ifndef JXL_CODESTREAM_HEADER_H_
#define JXL_CODESTREAM_HEADER_H_
#include <jxl/types.h>
typedef enum {
  JXL_ORIENT_IDENTITY = 1, JXL_ORIENT_FLIP_HORIZONTAL = 2, JXL_ORIENT_ROTATE_180 = 3, JXL_ORIENT_FLIP_VERTICAL = 4,
  JXL_ORIENT_TRANSPOSE = 5, JXL_ORIENT_ROTATE_90_CW = 6, JXL_ORIENT_ANTI_TRANSPOSE = 7, JXL_ORIENT_ROTATE_90_CCW = 8
} JxlOrientation;
typedef enum {
  JXL_CHANNEL_ALPHA, JXL_CHANNEL_DEPTH, JXL_CHANNEL_SPOT_COLOR, JXL_CHANNEL_SELECTION_MASK, JXL_CHANNEL_BLACK,
  JXL_CHANNEL_CFA, JXL_CHANNEL_THERMAL, JXL_CHANNEL_RESERVED0, JXL_CHANNEL_RESERVED1, JXL_CHANNEL_RESERVED2,
  JXL_CHANNEL_RESERVED3, JXL_CHANNEL_RESERVED4, JXL_CHANNEL_RESERVED5, JXL_CHANNEL_RESERVED6, JXL_CHANNEL_RESERVED7,
  JXL_CHANNEL_UNKNOWN, JXL_CHANNEL_OPTIONAL
} JxlExtraChannelType;
typedef struct { uint32_t xsize, ysize; } JxlPreviewHeader;
typedef struct { uint32_t tps_numerator, tps_denominator, num_loops; JXL_BOOL have_timecodes; } JxlAnimationHeader;
typedef struct {
  JXL_BOOL have_container;
  uint32_t xsize, ysize, bits_per_sample, exponent_bits_per_sample;
  float intensity_target, min_nits;
  JXL_BOOL relative_to_max_display;
  float linear_below;
  JXL_BOOL uses_original_profile, have_preview, have_animation;
  JxlOrientation orientation;
  uint32_t num_color_channels, num_extra_channels, alpha_bits, alpha_exponent_bits;
  JXL_BOOL alpha_premultiplied;
  JxlPreviewHeader preview;
  JxlAnimationHeader animation;
  uint32_t intrinsic_xsize, intrinsic_ysize;
  uint8_t padding[100];
} JxlBasicInfo;
typedef struct {
  JxlExtraChannelType type;
  uint32_t bits_per_sample, exponent_bits_per_sample, dim_shift, name_length;
  JXL_BOOL alpha_premultiplied;
  float spot_color[4];
  uint32_t cfa_channel;
} JxlExtraChannelInfo;
typedef enum { JXL_BLEND_REPLACE = 0, JXL_BLEND_ADD = 1, JXL_BLEND_BLEND = 2, JXL_BLEND_MULADD = 3, JXL_BLEND_MUL = 4 } JxlBlendMode;
typedef struct { JxlBlendMode blendmode; uint32_t source, alpha; JXL_BOOL clamp; } JxlBlendInfo;
typedef struct {
  JXL_BOOL have_crop;
  int32_t crop_x0, crop_y0;
  uint32_t xsize, ysize;
  JxlBlendInfo blend_info;
  uint32_t save_as_reference;
} JxlLayerInfo;
typedef struct {
  uint32_t duration, timecode, name_length;
  JXL_BOOL is_last;
  JxlLayerInfo layer_info;
} JxlFrameHeader;
#endif
