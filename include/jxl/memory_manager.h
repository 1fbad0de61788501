/* libjxl_amd: custom allocator hook (reference lib/include/jxl/memory_manager.h:28-65). */
#ifndef JXL_MEMORY_MANAGER_H_
#define JXL_MEMORY_MANAGER_H_
#include <stddef.h>
typedef void* (*jpegxl_alloc_func)(void* opaque, size_t size);
typedef void (*jpegxl_free_func)(void* opaque, void* address);
typedef struct JxlMemoryManagerStruct {
  void* opaque;
  jpegxl_alloc_func alloc;
  jpegxl_free_func free;
} JxlMemoryManager;
#endif
