// wrapper.h -- the one extern "C" entry of the reference tree (rustbind/wrapper.h:11-12): bindgen cannot return a std::shared_ptr, so
// the pool handle is written through a pointer.
#pragma once
#include "troy.h"

namespace troy_wrapper {
extern "C" void create_memory_pool_handle(size_t device_index, troy::MemoryPoolHandle* out);
}
