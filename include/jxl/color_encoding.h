/* libjxl_amd: colour description structs (layout as reference lib/include/jxl/color_encoding.h:24-155). */
#ifndef JXL_COLOR_ENCODING_H_
#define JXL_COLOR_ENCODING_H_
typedef enum { JXL_COLOR_SPACE_RGB, JXL_COLOR_SPACE_GRAY, JXL_COLOR_SPACE_XYB, JXL_COLOR_SPACE_UNKNOWN } JxlColorSpace;
typedef enum { JXL_WHITE_POINT_D65 = 1, JXL_WHITE_POINT_CUSTOM = 2, JXL_WHITE_POINT_E = 10, JXL_WHITE_POINT_DCI = 11 } JxlWhitePoint;
typedef enum { JXL_PRIMARIES_SRGB = 1, JXL_PRIMARIES_CUSTOM = 2, JXL_PRIMARIES_2100 = 9, JXL_PRIMARIES_P3 = 11 } JxlPrimaries;
typedef enum {
  JXL_TRANSFER_FUNCTION_709 = 1, JXL_TRANSFER_FUNCTION_UNKNOWN = 2, JXL_TRANSFER_FUNCTION_LINEAR = 8,
  JXL_TRANSFER_FUNCTION_SRGB = 13, JXL_TRANSFER_FUNCTION_PQ = 16, JXL_TRANSFER_FUNCTION_DCI = 17,
  JXL_TRANSFER_FUNCTION_HLG = 18, JXL_TRANSFER_FUNCTION_GAMMA = 65535
} JxlTransferFunction;
typedef enum {
  JXL_RENDERING_INTENT_PERCEPTUAL = 0, JXL_RENDERING_INTENT_RELATIVE, JXL_RENDERING_INTENT_SATURATION,
  JXL_RENDERING_INTENT_ABSOLUTE
} JxlRenderingIntent;
typedef struct {
  JxlColorSpace color_space;
  JxlWhitePoint white_point;
  double white_point_xy[2];
  JxlPrimaries primaries;
  double primaries_red_xy[2];
  double primaries_green_xy[2];
  double primaries_blue_xy[2];
  JxlTransferFunction transfer_function;
  double gamma;
  JxlRenderingIntent rendering_intent;
} JxlColorEncoding;
#endif
