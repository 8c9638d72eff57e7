// TEST INFRASTRUCTURE: shadows torch's header for the host build of the reference sources.
#ifndef SDN_REF_SHIM_CUDACONTEXT_H
#define SDN_REF_SHIM_CUDACONTEXT_H
#include "../../cuda_runtime.h"
namespace at { namespace cuda {
static inline cudaStream_t getCurrentCUDAStream(int = 0) { return nullptr; }
}}
#endif
