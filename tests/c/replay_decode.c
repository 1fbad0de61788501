/* Test program (plain C, compiled by the test suite against libjxl_amd.so): replays the JxlDecoder call sequence of the
 * reference's DecodeImageJXL (lib/extras/dec/jxl.cc:140-669) -- the glue under djxl and benchmark_xl -- including the
 * calls that are unconditional there: SetCms, Set/ReleaseBoxBuffer, GetBoxType, Set/ReleaseJPEGBuffer. It shows that the
 * library links at that level and produces the same event order.
 * usage: replay_decode IN.jxl OUT.raw {u8|u16|f16|f32} CHANNELS [callback|mt] [chunk=N] [linear]
 * Prints one line per event; exit code 0 = decoded, 3 = stopped at the pixels because no GPU is present, else failure. */
#include <jxl/decode.h>
#include <jxl/resizable_parallel_runner.h>
#include <jxl/thread_parallel_runner.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static uint8_t* g_pixels;
static size_t g_stride, g_bpp;
static void OnPixels(void* opaque, size_t x, size_t y, size_t n, const void* px) {
  (void)opaque;
  memcpy(g_pixels + y * g_stride + x * g_bpp, px, n * g_bpp);
}
static int g_mt_inits, g_mt_destroys;
static void* MtInit(void* opaque, size_t num_threads, size_t num_pixels) {
  (void)num_threads;
  (void)num_pixels;
  g_mt_inits++;
  return opaque;
}
static void MtRun(void* opaque, size_t thread, size_t x, size_t y, size_t n, const void* px) {
  (void)thread;
  OnPixels(opaque, x, y, n, px);
}
static void MtDestroy(void* opaque) {
  (void)opaque;
  g_mt_destroys++;
}
static size_t g_allocs, g_frees;
static void* CountingAlloc(void* opaque, size_t n) {
  (void)opaque;
  g_allocs++;
  return malloc(n);
}
static void CountingFree(void* opaque, void* p) {
  (void)opaque;
  if (p) g_frees++;
  free(p);
}

int main(int argc, char** argv) {
  if (argc < 5) return 2;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 2;
  fseek(f, 0, SEEK_END);
  size_t size = (size_t)ftell(f);
  fseek(f, 0, SEEK_SET);
  uint8_t* bytes = (uint8_t*)malloc(size);
  if (fread(bytes, 1, size, f) != size) return 2;
  fclose(f);
  JxlPixelFormat format = {(uint32_t)atoi(argv[4]), JXL_TYPE_UINT8, JXL_NATIVE_ENDIAN, 0};
  if (!strcmp(argv[3], "u16")) format.data_type = JXL_TYPE_UINT16;
  if (!strcmp(argv[3], "f16")) format.data_type = JXL_TYPE_FLOAT16;
  if (!strcmp(argv[3], "f32")) format.data_type = JXL_TYPE_FLOAT;
  int use_callback = 0, use_mt = 0, linear = 0, keep = 0, want_ec = 0, multi = 0, frames_done = 0, swap = 0, layers = 0;
  uint32_t cur_w = 0, cur_h = 0; /* size of the frame being delivered: the image, or (layers) the frame's own */
  uint8_t* preview_pixels = NULL;
  size_t preview_bytes = 0;
  int unpremul = 0, progression = 0, detail = -1;
  int nopreview = 0; /* nopreview: do not subscribe to JXL_DEC_PREVIEW_IMAGE (the preview frame is stepped over) */
  int flush = 0, in_frame = 0, flushes = 0; /* flush: JxlDecoderFlushImage whenever the decoder runs out of input inside a frame */
  size_t skip = 0;
  long skipcur = -1; /* skipcur=K: JxlDecoderSkipCurrentFrame at the FRAME event of the K-th frame that has one */
  long frame_events = 0;
  uint8_t* ec_pixels[4] = {NULL, NULL, NULL, NULL};
  size_t ec_sizes[4] = {0, 0, 0, 0};
  size_t chunk = 0;
  for (int i = 5; i < argc; i++) {
    if (!strcmp(argv[i], "callback")) use_callback = 1;
    if (!strcmp(argv[i], "mt")) use_mt = 1;
    if (!strcmp(argv[i], "linear")) linear = 1;
    if (!strcmp(argv[i], "swap")) swap = 1; /* hand the input back and in again after every frame (decode.h: JxlDecoderReleaseInput) */
    if (!strcmp(argv[i], "frames")) multi = 1; /* animation: every frame's pixels are appended to the output file */
    if (!strcmp(argv[i], "layers")) layers = multi = 1; /* JxlDecoderSetCoalescing(false): every regular frame by itself */
    if (!strncmp(argv[i], "skip=", 5)) skip = (size_t)atol(argv[i] + 5); /* JxlDecoderSkipFrames before decoding */
    if (!strncmp(argv[i], "skipcur=", 8)) skipcur = atol(argv[i] + 8);
    if (!strcmp(argv[i], "keep")) keep = 1; /* the pixels as coded, the orientation left to the caller */
    if (!strcmp(argv[i], "ec")) want_ec = 1; /* also fetch every extra channel into its own buffer (jxl.cc:571-590) */
    if (!strncmp(argv[i], "chunk=", 6)) chunk = (size_t)atol(argv[i] + 6);
    if (!strcmp(argv[i], "flush")) flush = 1;
    if (!strcmp(argv[i], "nopreview")) nopreview = 1;
    if (!strcmp(argv[i], "unpremul")) unpremul = 1; /* JxlDecoderSetUnpremultiplyAlpha(true) */
    if (!strcmp(argv[i], "progression")) progression = 1; /* subscribe to JXL_DEC_FRAME_PROGRESSION and flush when it comes */
    if (!strncmp(argv[i], "detail=", 7)) detail = atoi(argv[i] + 7); /* JxlDecoderSetProgressiveDetail */
  }
  if (JxlSignatureCheck(bytes, size) == JXL_SIG_INVALID) return 2;
  JxlMemoryManager mm = {NULL, CountingAlloc, CountingFree};
  JxlDecoder* dec = JxlDecoderCreate(&mm);
  void* runner = JxlThreadParallelRunnerCreate(NULL, 4);
  if (!dec || !runner) return 2;
  if (JxlDecoderSetParallelRunner(dec, JxlThreadParallelRunner, runner) != JXL_DEC_SUCCESS) return 2;
  int events = JXL_DEC_BASIC_INFO | JXL_DEC_FULL_IMAGE | JXL_DEC_COLOR_ENCODING | JXL_DEC_FRAME | JXL_DEC_PREVIEW_IMAGE | JXL_DEC_BOX;
  if (nopreview) events &= ~JXL_DEC_PREVIEW_IMAGE;
  if (progression) events |= JXL_DEC_FRAME_PROGRESSION;
  if (JxlDecoderSubscribeEvents(dec, events) != JXL_DEC_SUCCESS) return 2;
  if (JxlDecoderSetRenderSpotcolors(dec, JXL_TRUE) != JXL_DEC_SUCCESS) return 2;
  if (JxlDecoderSetKeepOrientation(dec, keep ? JXL_TRUE : JXL_FALSE) != JXL_DEC_SUCCESS) return 2;
  if (JxlDecoderSetUnpremultiplyAlpha(dec, unpremul ? JXL_TRUE : JXL_FALSE) != JXL_DEC_SUCCESS) return 2;
  if (JxlDecoderSetCoalescing(dec, layers ? JXL_FALSE : JXL_TRUE) != JXL_DEC_SUCCESS) return 2;
  if (detail >= 0 && JxlDecoderSetProgressiveDetail(dec, (JxlProgressiveDetail)detail) != JXL_DEC_SUCCESS) return 2;
  if (JxlDecoderSetDecompressBoxes(dec, JXL_TRUE) != JXL_DEC_SUCCESS) return 2;
  if (skip) JxlDecoderSkipFrames(dec, skip);
  /* input in one piece, or in chunks the way a streaming caller feeds it (decode.h: unprocessed bytes are re-supplied) */
  size_t given = chunk && chunk < size ? chunk : size, consumed = 0;
  if (JxlDecoderSetInput(dec, bytes, given) != JXL_DEC_SUCCESS) return 2;
  if (given == size) JxlDecoderCloseInput(dec);
  JxlBasicInfo info;
  JxlFrameHeader fh;
  memset(&info, 0, sizeof(info));
  uint8_t box_buffer[64];
  int have_box_buffer = 0, rc = 1;
  for (;;) {
    JxlDecoderStatus st = JxlDecoderProcessInput(dec);
    if (st == JXL_DEC_ERROR) {
      printf("event ERROR\n");
      {
        extern const char* jxlamd_last_error(void);
        printf("last error: %s\n", jxlamd_last_error());
        /* 3 = everything up to the pixels worked and there is no GPU to make them (the CPU-only test box) */
        rc = JxlDecoderGetFrameHeader(dec, &fh) == JXL_DEC_SUCCESS && g_pixels && strstr(jxlamd_last_error(), "no HIP device") ? 3 : 1;
      }
      break;
    } else if (st == JXL_DEC_NEED_MORE_INPUT) {
      if (flush && in_frame) { /* decode.h: between the frame's output buffer and its FULL_IMAGE */
        if (JxlDecoderFlushImage(dec) == JXL_DEC_SUCCESS) {
          char name[1024];
          FILE* o;
          snprintf(name, sizeof(name), "%s.flush%d", argv[2], flushes);
          o = fopen(name, "wb");
          fwrite(g_pixels, 1, g_stride * cur_h, o);
          fclose(o);
          printf("flushed %d bytes_given=%zu\n", flushes++, consumed + given);
        } else {
          {
          extern const char* jxlamd_last_error(void);
          printf("flush refused bytes_given=%zu (%s)\n", consumed + given, jxlamd_last_error());
        }
        }
      }
      size_t left = JxlDecoderReleaseInput(dec);
      consumed += given - left;
      if (consumed + left >= size) {
        printf("event NEED_MORE_INPUT at end of file\n");
        break;
      }
      given = size - consumed < left + chunk ? size - consumed : left + chunk;
      printf("event NEED_MORE_INPUT consumed=%zu\n", consumed);
      if (JxlDecoderSetInput(dec, bytes + consumed, given) != JXL_DEC_SUCCESS) return 2;
      if (consumed + given == size) JxlDecoderCloseInput(dec);
    } else if (st == JXL_DEC_FRAME_PROGRESSION) {
      printf("event FRAME_PROGRESSION ratio=%zu bytes_given=%zu\n", JxlDecoderGetIntendedDownsamplingRatio(dec), consumed + given);
      if (JxlDecoderFlushImage(dec) == JXL_DEC_SUCCESS) {
        char name[1024];
        FILE* o;
        snprintf(name, sizeof(name), "%s.flush%d", argv[2], flushes);
        o = fopen(name, "wb");
        fwrite(g_pixels, 1, g_stride * cur_h, o);
        fclose(o);
        printf("flushed %d bytes_given=%zu\n", flushes++, consumed + given);
      } else {
        {
          extern const char* jxlamd_last_error(void);
          printf("flush refused bytes_given=%zu (%s)\n", consumed + given, jxlamd_last_error());
        }
      }
    } else if (st == JXL_DEC_BOX) {
      JxlBoxType type;
      if (have_box_buffer) JxlDecoderReleaseBoxBuffer(dec);
      have_box_buffer = 0;
      if (JxlDecoderGetBoxType(dec, type, JXL_TRUE) != JXL_DEC_SUCCESS) return 2;
      printf("event BOX %.4s\n", type);
      if (memcmp(type, "jxlc", 4) && memcmp(type, "jxlp", 4)) {
        if (JxlDecoderSetBoxBuffer(dec, box_buffer, sizeof(box_buffer)) != JXL_DEC_SUCCESS) return 2;
        have_box_buffer = 1;
      }
    } else if (st == JXL_DEC_BOX_NEED_MORE_OUTPUT) {
      JxlDecoderReleaseBoxBuffer(dec);
      if (JxlDecoderSetBoxBuffer(dec, box_buffer, sizeof(box_buffer)) != JXL_DEC_SUCCESS) return 2;
    } else if (st == JXL_DEC_JPEG_RECONSTRUCTION) {
      if (JxlDecoderSetJPEGBuffer(dec, box_buffer, sizeof(box_buffer)) != JXL_DEC_SUCCESS) return 2;
    } else if (st == JXL_DEC_BASIC_INFO) {
      if (JxlDecoderGetBasicInfo(dec, &info) != JXL_DEC_SUCCESS) return 2;
      printf("event BASIC_INFO %ux%u bits=%u extra=%u alpha_bits=%u container=%d\n", info.xsize, info.ysize, info.bits_per_sample,
             info.num_extra_channels, info.alpha_bits, info.have_container);
      printf("orientation=%d\n", (int)info.orientation);
      for (uint32_t i = 0; i < info.num_extra_channels; i++) {
        JxlExtraChannelInfo eci;
        char name[8];
        if (JxlDecoderGetExtraChannelInfo(dec, i, &eci) != JXL_DEC_SUCCESS) return 2;
        if (JxlDecoderGetExtraChannelName(dec, i, name, sizeof(name)) != JXL_DEC_SUCCESS) return 2;
        printf("extra channel %u type=%d bits=%u\n", i, (int)eci.type, eci.bits_per_sample);
      }
    } else if (st == JXL_DEC_COLOR_ENCODING) {
      JxlColorEncoding ce;
      JxlCmsInterface cms;
      size_t icc_size = 1;
      memset(&cms, 0, sizeof(cms));
      if (JxlDecoderSetCms(dec, cms) != JXL_DEC_SUCCESS) return 2; /* jxl.cc:406 */
      memset(&ce, 0, sizeof(ce));
      ce.transfer_function = (JxlTransferFunction)-1;
      /* (jxl.cc:430-440 tolerates images whose pixels only have an ICC description) */
      if (JxlDecoderGetColorAsEncodedProfile(dec, JXL_COLOR_PROFILE_TARGET_DATA, &ce) != JXL_DEC_SUCCESS && linear) return 2;
      if (linear) {
        ce.transfer_function = JXL_TRANSFER_FUNCTION_LINEAR;
        if (JxlDecoderSetOutputColorProfile(dec, &ce, NULL, 0) != JXL_DEC_SUCCESS) return 2;
        if (JxlDecoderGetColorAsEncodedProfile(dec, JXL_COLOR_PROFILE_TARGET_DATA, &ce) != JXL_DEC_SUCCESS) return 2;
      }
      if (JxlDecoderGetICCProfileSize(dec, JXL_COLOR_PROFILE_TARGET_DATA, &icc_size) != JXL_DEC_SUCCESS) icc_size = 0;
      printf("event COLOR_ENCODING tf=%d icc=%zu\n", (int)ce.transfer_function, icc_size);
      {
        /* the original profile, as jxl.cc:411-428 fetches it */
        size_t orig = 0;
        if (JxlDecoderGetICCProfileSize(dec, JXL_COLOR_PROFILE_TARGET_ORIGINAL, &orig) == JXL_DEC_SUCCESS && orig) {
          unsigned char* icc = (unsigned char*)malloc(orig);
          unsigned long long h = 1469598103934665603ull;
          size_t i;
          if (!icc || JxlDecoderGetColorAsICCProfile(dec, JXL_COLOR_PROFILE_TARGET_ORIGINAL, icc, orig) != JXL_DEC_SUCCESS) return 2;
          for (i = 0; i < orig; i++) h = (h ^ icc[i]) * 1099511628211ull;
          printf("original icc size=%zu fnv1a=%016llx encoded_profile=%d\n", orig, h,
                 JxlDecoderGetColorAsEncodedProfile(dec, JXL_COLOR_PROFILE_TARGET_ORIGINAL, NULL) == JXL_DEC_SUCCESS);
          free(icc);
        }
      }
    } else if (st == JXL_DEC_FRAME) {
      char name[4];
      if (JxlDecoderGetFrameHeader(dec, &fh) != JXL_DEC_SUCCESS) return 2;
      if (JxlDecoderGetFrameName(dec, name, sizeof(name)) != JXL_DEC_SUCCESS) return 2;
      printf("event FRAME %ux%u last=%d downsampling=%zu\n", fh.layer_info.xsize, fh.layer_info.ysize, fh.is_last,
             JxlDecoderGetIntendedDownsamplingRatio(dec));
      cur_w = layers ? fh.layer_info.xsize : info.xsize;
      cur_h = layers ? fh.layer_info.ysize : info.ysize;
      if (layers)
        printf("layer crop=%d x0=%d y0=%d blend=%d source=%u alpha=%u clamp=%d save_as=%u\n", fh.layer_info.have_crop, fh.layer_info.crop_x0,
               fh.layer_info.crop_y0, (int)fh.layer_info.blend_info.blendmode, fh.layer_info.blend_info.source, fh.layer_info.blend_info.alpha,
               fh.layer_info.blend_info.clamp, fh.layer_info.save_as_reference);
      if (info.have_animation)
        printf("animation tps=%u/%u loops=%u duration=%u timecode=%u\n", info.animation.tps_numerator, info.animation.tps_denominator,
               info.animation.num_loops, fh.duration, fh.timecode);
      if (frame_events++ == skipcur) {
        if (JxlDecoderSkipCurrentFrame(dec) != JXL_DEC_SUCCESS) return 2;
        printf("skipped current frame\n");
      }
    } else if (st == JXL_DEC_NEED_IMAGE_OUT_BUFFER) {
      size_t buffer_size = 0;
      if (JxlDecoderImageOutBufferSize(dec, &format, &buffer_size) != JXL_DEC_SUCCESS) return 2;
      g_bpp = format.num_channels * (format.data_type == JXL_TYPE_UINT8 ? 1 : (format.data_type == JXL_TYPE_FLOAT ? 4 : 2));
      if (!cur_w) { cur_w = info.xsize; cur_h = info.ysize; } /* (no FRAME event subscribed / delivered) */
      g_stride = (size_t)cur_w * g_bpp;
      if (buffer_size != g_stride * cur_h) return 2;
      free(g_pixels);
      g_pixels = (uint8_t*)calloc(buffer_size, 1);
      printf("event NEED_IMAGE_OUT_BUFFER size=%zu\n", buffer_size);
      in_frame = 1;
      if (use_mt) {
        if (JxlDecoderSetMultithreadedImageOutCallback(dec, &format, MtInit, MtRun, MtDestroy, &info) != JXL_DEC_SUCCESS) return 2;
      } else if (use_callback) {
        if (JxlDecoderSetImageOutCallback(dec, &format, OnPixels, NULL) != JXL_DEC_SUCCESS) return 2;
      } else if (JxlDecoderSetImageOutBuffer(dec, &format, g_pixels, buffer_size) != JXL_DEC_SUCCESS) {
        return 2;
      }
      JxlBitDepth depth = {JXL_BIT_DEPTH_FROM_PIXEL_FORMAT, 0, 0};
      if (JxlDecoderSetImageOutBitDepth(dec, &depth) != JXL_DEC_SUCCESS) return 2; /* jxl.cc:567 */
      for (uint32_t i = 0; i < info.num_extra_channels; i++) {
        size_t ec_size = 0;
        if (JxlDecoderExtraChannelBufferSize(dec, &format, &ec_size, i) != JXL_DEC_SUCCESS) return 2;
        if (want_ec && i < 4) {
          if (ec_size != (size_t)cur_w * cur_h * (g_bpp / format.num_channels)) return 2;
          ec_pixels[i] = (uint8_t*)calloc(ec_size, 1);
          ec_sizes[i] = ec_size;
          if (JxlDecoderSetExtraChannelBuffer(dec, &format, ec_pixels[i], ec_size, i) != JXL_DEC_SUCCESS) return 2;
        }
      }
    } else if (st == JXL_DEC_NEED_PREVIEW_OUT_BUFFER) { /* jxl.cc:300-320: the preview frame has a buffer of its own */
      size_t preview_size = 0;
      if (JxlDecoderPreviewOutBufferSize(dec, &format, &preview_size) != JXL_DEC_SUCCESS) return 2;
      free(preview_pixels);
      preview_pixels = (uint8_t*)calloc(preview_size, 1);
      preview_bytes = preview_size;
      printf("event NEED_PREVIEW_OUT_BUFFER size=%zu preview=%ux%u\n", preview_size, info.preview.xsize, info.preview.ysize);
      if (JxlDecoderSetPreviewOutBuffer(dec, &format, preview_pixels, preview_size) != JXL_DEC_SUCCESS) return 2;
    } else if (st == JXL_DEC_PREVIEW_IMAGE) {
      char name[1024];
      FILE* o;
      printf("event PREVIEW_IMAGE\n");
      snprintf(name, sizeof(name), "%s.preview", argv[2]);
      o = fopen(name, "wb");
      fwrite(preview_pixels, 1, preview_bytes, o);
      fclose(o);
    } else if (st == JXL_DEC_FULL_IMAGE) {
      printf("event FULL_IMAGE\n");
      in_frame = 0;
      if (multi && g_pixels) {
        FILE* o = fopen(argv[2], frames_done ? "ab" : "wb");
        fwrite(g_pixels, 1, g_stride * cur_h, o);
        fclose(o);
        frames_done++;
      }
      if (swap) {
        size_t left = JxlDecoderReleaseInput(dec);
        consumed += given - left;
        given = size - consumed;
        printf("swap consumed=%zu\n", consumed);
        if (given) {
          if (JxlDecoderSetInput(dec, bytes + consumed, given) != JXL_DEC_SUCCESS) return 2;
          JxlDecoderCloseInput(dec);
        }
      }
    } else if (st == JXL_DEC_SUCCESS) {
      printf("event SUCCESS\n");
      rc = 0;
      break;
    } else {
      printf("event %d\n", (int)st);
      break;
    }
  }
  if (have_box_buffer) JxlDecoderReleaseBoxBuffer(dec);
  JxlDecoderReleaseJPEGBuffer(dec); /* jxl.cc:658 */
  printf("unprocessed=%zu\n", JxlDecoderReleaseInput(dec));
  if (rc == 0 && g_pixels && !multi) {
    FILE* o = fopen(argv[2], "wb");
    fwrite(g_pixels, 1, g_stride * info.ysize, o);
    for (int i = 0; i < 4; i++) /* the extra channel planes follow the pixels */
      if (ec_pixels[i]) fwrite(ec_pixels[i], 1, ec_sizes[i], o);
    fclose(o);
  }
  free(preview_pixels);
  if (use_mt) printf("mt init=%d destroy=%d\n", g_mt_inits, g_mt_destroys);
  JxlDecoderDestroy(dec);
  JxlThreadParallelRunnerDestroy(runner);
  printf("memory manager: allocs=%zu frees=%zu\n", g_allocs, g_frees);
  return g_allocs == g_frees && g_allocs > 0 ? rc : 4;
}
