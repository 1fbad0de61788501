/* libjxl_amd: the JxlParallelRunner hook (reference lib/include/jxl/parallel_runner.h:59-121). The library never
 * creates threads; it calls runner(runner_opaque, jpegxl_opaque, init, func, begin, end). */
#ifndef JXL_PARALLEL_RUNNER_H_
#define JXL_PARALLEL_RUNNER_H_
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef int JxlParallelRetCode;
#define JXL_PARALLEL_RET_SUCCESS (0)
#define JXL_PARALLEL_RET_RUNNER_ERROR (-1)
typedef JxlParallelRetCode (*JxlParallelRunInit)(void* jpegxl_opaque, size_t num_threads);
typedef void (*JxlParallelRunFunction)(void* jpegxl_opaque, uint32_t value, size_t thread_id);
typedef JxlParallelRetCode (*JxlParallelRunner)(void* runner_opaque, void* jpegxl_opaque, JxlParallelRunInit init,
                                                JxlParallelRunFunction func, uint32_t start_range, uint32_t end_range);
#ifdef __cplusplus
}
#endif
#endif
