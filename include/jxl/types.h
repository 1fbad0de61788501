/* libjxl_amd: ABI-compatible subset of the JPEG XL C API types (layout as in reference lib/include/jxl/types.h:23-150). */
#ifndef JXL_TYPES_H_
#define JXL_TYPES_H_
#include <stddef.h>
#include <stdint.h>
#define JXL_BOOL int
#define JXL_TRUE 1
#define JXL_FALSE 0
#define JXL_EXPORT __attribute__((visibility("default")))
#define JXL_THREADS_EXPORT __attribute__((visibility("default")))
typedef enum { JXL_TYPE_FLOAT = 0, JXL_TYPE_UINT8 = 2, JXL_TYPE_UINT16 = 3, JXL_TYPE_FLOAT16 = 5 } JxlDataType;
typedef enum { JXL_NATIVE_ENDIAN = 0, JXL_LITTLE_ENDIAN = 1, JXL_BIG_ENDIAN = 2 } JxlEndianness;
typedef struct {
  uint32_t num_channels;
  JxlDataType data_type;
  JxlEndianness endianness;
  size_t align;
} JxlPixelFormat;
typedef enum { JXL_BIT_DEPTH_FROM_PIXEL_FORMAT = 0, JXL_BIT_DEPTH_FROM_CODESTREAM = 1, JXL_BIT_DEPTH_CUSTOM = 2 } JxlBitDepthType;
typedef struct {
  JxlBitDepthType type;
  uint32_t bits_per_sample;
  uint32_t exponent_bits_per_sample;
} JxlBitDepth;
typedef char JxlBoxType[4];
#endif
