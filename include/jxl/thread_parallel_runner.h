/* libjxl_amd: std::thread implementation of JxlParallelRunner (reference lib/include/jxl/thread_parallel_runner.h,
 * lib/threads/thread_parallel_runner.cc:67-109). */
#ifndef JXL_THREAD_PARALLEL_RUNNER_H_
#define JXL_THREAD_PARALLEL_RUNNER_H_
#include <jxl/memory_manager.h>
#include <jxl/parallel_runner.h>
#include <jxl/types.h>
#ifdef __cplusplus
extern "C" {
#endif
JXL_THREADS_EXPORT JxlParallelRetCode JxlThreadParallelRunner(void* runner_opaque, void* jpegxl_opaque,
                                                              JxlParallelRunInit init, JxlParallelRunFunction func,
                                                              uint32_t start_range, uint32_t end_range);
JXL_THREADS_EXPORT void* JxlThreadParallelRunnerCreate(const JxlMemoryManager* memory_manager, size_t num_worker_threads);
JXL_THREADS_EXPORT void JxlThreadParallelRunnerDestroy(void* runner_opaque);
JXL_THREADS_EXPORT size_t JxlThreadParallelRunnerDefaultNumWorkerThreads(void);
#ifdef __cplusplus
}
#endif
#endif
