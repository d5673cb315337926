/* Controls of the stand-in HIP runtime (tests/hip_stub/hip_stub.cpp) -- test infrastructure only. */
#pragma once
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
/* the n-th allocation (hipMalloc / hipHostMalloc) from now fails, n = 0 the next one; -1: never */
void hip_stub_fail_alloc_after(long n);
unsigned long long hip_stub_launches(void);
unsigned long long hip_stub_copies(void);
size_t hip_stub_live_allocations(void);
/* run fn(arg) on the worker of `stream`, in stream order (the RCCL stand-in queues its collectives with this) */
int hip_stub_enqueue(void* stream, void (*fn)(void*), void* arg);
int hip_stub_current_device(void);
#ifdef __cplusplus
}
#endif
