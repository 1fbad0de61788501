/* libjxl_amd: the JxlDecoder C API subset that DecodeImageJXL (reference lib/extras/dec/jxl.cc:140-669) and
 * djxl/benchmark_xl drive. Same names, argument meaning and status values as reference lib/include/jxl/decode.h;
 * the entry points cite the reference implementation they replace (lib/jxl/decode.cc). The decode itself runs on
 * the MI355X through include/jxl_amd_hip.h; there is no CPU pixel path behind this API. */
#ifndef JXL_DECODE_H_
#define JXL_DECODE_H_
#include <jxl/cms_interface.h>
#include <jxl/codestream_header.h>
#include <jxl/color_encoding.h>
#include <jxl/memory_manager.h>
#include <jxl/parallel_runner.h>
#include <jxl/types.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef enum { JXL_SIG_NOT_ENOUGH_BYTES = 0, JXL_SIG_INVALID = 1, JXL_SIG_CODESTREAM = 2, JXL_SIG_CONTAINER = 3 } JxlSignature;
typedef enum {
  JXL_DEC_SUCCESS = 0, JXL_DEC_ERROR = 1, JXL_DEC_NEED_MORE_INPUT = 2, JXL_DEC_NEED_PREVIEW_OUT_BUFFER = 3,
  JXL_DEC_NEED_IMAGE_OUT_BUFFER = 5, JXL_DEC_JPEG_NEED_MORE_OUTPUT = 6, JXL_DEC_BOX_NEED_MORE_OUTPUT = 7,
  JXL_DEC_BASIC_INFO = 0x40, JXL_DEC_COLOR_ENCODING = 0x100, JXL_DEC_PREVIEW_IMAGE = 0x200, JXL_DEC_FRAME = 0x400,
  JXL_DEC_FULL_IMAGE = 0x1000, JXL_DEC_JPEG_RECONSTRUCTION = 0x2000, JXL_DEC_BOX = 0x4000,
  JXL_DEC_FRAME_PROGRESSION = 0x8000, JXL_DEC_BOX_COMPLETE = 0x10000
} JxlDecoderStatus;
typedef enum { JXL_COLOR_PROFILE_TARGET_ORIGINAL = 0, JXL_COLOR_PROFILE_TARGET_DATA = 1 } JxlColorProfileTarget;
typedef enum { kFrames = 0, kDC = 1, kLastPasses = 2, kPasses = 3, kDCProgressive = 4, kDCGroups = 5, kGroups = 6 } JxlProgressiveDetail;
typedef struct JxlDecoderStruct JxlDecoder;
typedef void (*JxlImageOutCallback)(void* opaque, size_t x, size_t y, size_t num_pixels, const void* pixels);
/* Multi-threaded form (decode.h:1055-1085): init returns a run_opaque, run delivers pixel runs, destroy ends it. */
typedef void* (*JxlImageOutInitCallback)(void* init_opaque, size_t num_threads, size_t num_pixels_per_thread);
typedef void (*JxlImageOutRunCallback)(void* run_opaque, size_t thread_id, size_t x, size_t y, size_t num_pixels,
                                       const void* pixels);
typedef void (*JxlImageOutDestroyCallback)(void* run_opaque);

JXL_EXPORT uint32_t JxlDecoderVersion(void);                                            /* decode.cc:151 */
JXL_EXPORT JxlSignature JxlSignatureCheck(const uint8_t* buf, size_t len);              /* decode.cc:115-159 */
JXL_EXPORT JxlDecoder* JxlDecoderCreate(const JxlMemoryManager* memory_manager);        /* decode.cc:844 */
JXL_EXPORT void JxlDecoderReset(JxlDecoder* dec);                                       /* decode.cc:827 */
JXL_EXPORT void JxlDecoderDestroy(JxlDecoder* dec);                                     /* decode.cc:861 */
JXL_EXPORT void JxlDecoderRewind(JxlDecoder* dec);                                      /* decode.cc:870 */
JXL_EXPORT void JxlDecoderSkipFrames(JxlDecoder* dec, size_t amount);                   /* decode.cc:880 */
JXL_EXPORT JxlDecoderStatus JxlDecoderSkipCurrentFrame(JxlDecoder* dec);                /* decode.cc:897 */
JXL_EXPORT JxlDecoderStatus JxlDecoderSetParallelRunner(JxlDecoder* dec, JxlParallelRunner parallel_runner,
                                                        void* parallel_runner_opaque); /* decode.cc:918-927 */
JXL_EXPORT size_t JxlDecoderSizeHintBasicInfo(const JxlDecoder* dec);
JXL_EXPORT JxlDecoderStatus JxlDecoderSubscribeEvents(JxlDecoder* dec, int events_wanted); /* decode.cc:934 */
JXL_EXPORT JxlDecoderStatus JxlDecoderSetKeepOrientation(JxlDecoder* dec, JXL_BOOL skip_reorientation);
JXL_EXPORT JxlDecoderStatus JxlDecoderSetUnpremultiplyAlpha(JxlDecoder* dec, JXL_BOOL unpremul_alpha);
JXL_EXPORT JxlDecoderStatus JxlDecoderSetRenderSpotcolors(JxlDecoder* dec, JXL_BOOL render_spotcolors);
JXL_EXPORT JxlDecoderStatus JxlDecoderSetCoalescing(JxlDecoder* dec, JXL_BOOL coalescing);
JXL_EXPORT JxlDecoderStatus JxlDecoderProcessInput(JxlDecoder* dec);                    /* decode.cc:2159 */
JXL_EXPORT JxlDecoderStatus JxlDecoderSetInput(JxlDecoder* dec, const uint8_t* data, size_t size); /* :1574 */
JXL_EXPORT size_t JxlDecoderReleaseInput(JxlDecoder* dec);                              /* decode.cc:1586 */
JXL_EXPORT void JxlDecoderCloseInput(JxlDecoder* dec);                                  /* decode.cc:1595 */
JXL_EXPORT JxlDecoderStatus JxlDecoderGetBasicInfo(const JxlDecoder* dec, JxlBasicInfo* info); /* :2214 */
JXL_EXPORT JxlDecoderStatus JxlDecoderGetExtraChannelInfo(const JxlDecoder* dec, size_t index, JxlExtraChannelInfo* info);
JXL_EXPORT JxlDecoderStatus JxlDecoderGetExtraChannelName(const JxlDecoder* dec, size_t index, char* name, size_t size);
JXL_EXPORT JxlDecoderStatus JxlDecoderGetColorAsEncodedProfile(const JxlDecoder* dec, JxlColorProfileTarget target,
                                                               JxlColorEncoding* color_encoding); /* :2361 */
JXL_EXPORT JxlDecoderStatus JxlDecoderGetICCProfileSize(const JxlDecoder* dec, JxlColorProfileTarget target, size_t* size);
JXL_EXPORT JxlDecoderStatus JxlDecoderGetColorAsICCProfile(const JxlDecoder* dec, JxlColorProfileTarget target,
                                                           uint8_t* icc_profile, size_t size);
JXL_EXPORT JxlDecoderStatus JxlDecoderSetPreferredColorProfile(JxlDecoder* dec, const JxlColorEncoding* color_encoding);
JXL_EXPORT JxlDecoderStatus JxlDecoderSetDesiredIntensityTarget(JxlDecoder* dec, float desired_intensity_target);
JXL_EXPORT JxlDecoderStatus JxlDecoderSetOutputColorProfile(JxlDecoder* dec, const JxlColorEncoding* color_encoding,
                                                            const uint8_t* icc_data, size_t icc_size); /* :2810 */
JXL_EXPORT JxlDecoderStatus JxlDecoderGetFrameHeader(const JxlDecoder* dec, JxlFrameHeader* header); /* :2705 */
JXL_EXPORT JxlDecoderStatus JxlDecoderGetFrameName(const JxlDecoder* dec, char* name, size_t size); /* :2790 */
JXL_EXPORT JxlDecoderStatus JxlDecoderPreviewOutBufferSize(const JxlDecoder* dec, const JxlPixelFormat* format, size_t* size);
JXL_EXPORT JxlDecoderStatus JxlDecoderSetPreviewOutBuffer(JxlDecoder* dec, const JxlPixelFormat* format, void* buffer, size_t size);
JXL_EXPORT JxlDecoderStatus JxlDecoderImageOutBufferSize(const JxlDecoder* dec, const JxlPixelFormat* format,
                                                         size_t* size); /* decode.cc:2566 */
JXL_EXPORT JxlDecoderStatus JxlDecoderSetImageOutBuffer(JxlDecoder* dec, const JxlPixelFormat* format, void* buffer,
                                                        size_t size); /* decode.cc:2576-2606 */
JXL_EXPORT JxlDecoderStatus JxlDecoderSetImageOutCallback(JxlDecoder* dec, const JxlPixelFormat* format,
                                                          JxlImageOutCallback callback, void* opaque); /* :2651 */
JXL_EXPORT JxlDecoderStatus JxlDecoderExtraChannelBufferSize(const JxlDecoder* dec, const JxlPixelFormat* format,
                                                             size_t* size, uint32_t index);
JXL_EXPORT JxlDecoderStatus JxlDecoderSetExtraChannelBuffer(JxlDecoder* dec, const JxlPixelFormat* format, void* buffer,
                                                            size_t size, uint32_t index);
JXL_EXPORT JxlDecoderStatus JxlDecoderSetDecompressBoxes(JxlDecoder* dec, JXL_BOOL decompress);
JXL_EXPORT JxlDecoderStatus JxlDecoderSetProgressiveDetail(JxlDecoder* dec, JxlProgressiveDetail detail);
JXL_EXPORT size_t JxlDecoderGetIntendedDownsamplingRatio(JxlDecoder* dec);
JXL_EXPORT JxlDecoderStatus JxlDecoderFlushImage(JxlDecoder* dec);
JXL_EXPORT JxlDecoderStatus JxlDecoderSetImageOutBitDepth(JxlDecoder* dec, const JxlBitDepth* bit_depth); /* :2994 */
JXL_EXPORT JxlDecoderStatus JxlDecoderSetMultithreadedImageOutCallback(JxlDecoder* dec, const JxlPixelFormat* format,
                                                                       JxlImageOutInitCallback init_callback,
                                                                       JxlImageOutRunCallback run_callback,
                                                                       JxlImageOutDestroyCallback destroy_callback,
                                                                       void* init_opaque); /* decode.cc:2664-2703 */
/* Colour management (decode.cc:2480): accepted and kept, never called (sRGB / linear sRGB output only). */
JXL_EXPORT JxlDecoderStatus JxlDecoderSetCms(JxlDecoder* dec, JxlCmsInterface cms);
JXL_EXPORT JxlDecoderStatus JxlDecoderGetExtraChannelBlendInfo(const JxlDecoder* dec, size_t index, JxlBlendInfo* blend_info);
/* Container boxes (decode.cc:2852-2990). Box contents are handed out as stored: Brotli ("brob") boxes are not
 * decompressed (no Brotli in this build: a decompressed view of such a box is empty). */
JXL_EXPORT JxlDecoderStatus JxlDecoderSetBoxBuffer(JxlDecoder* dec, uint8_t* data, size_t size);
JXL_EXPORT size_t JxlDecoderReleaseBoxBuffer(JxlDecoder* dec);
JXL_EXPORT JxlDecoderStatus JxlDecoderGetBoxType(JxlDecoder* dec, JxlBoxType type, JXL_BOOL decompressed);
JXL_EXPORT JxlDecoderStatus JxlDecoderGetBoxSizeRaw(const JxlDecoder* dec, uint64_t* size);
JXL_EXPORT JxlDecoderStatus JxlDecoderGetBoxSizeContents(const JxlDecoder* dec, uint64_t* size);
/* JPEG reconstruction (decode.cc:2821-2850) is outside this path: the buffer calls fail, the event never occurs. */
JXL_EXPORT JxlDecoderStatus JxlDecoderSetJPEGBuffer(JxlDecoder* dec, uint8_t* data, size_t size);
JXL_EXPORT size_t JxlDecoderReleaseJPEGBuffer(JxlDecoder* dec);
#ifdef __cplusplus
}
#endif
#endif
