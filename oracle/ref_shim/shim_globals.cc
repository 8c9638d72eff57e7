// TEST INFRASTRUCTURE: storage for the per-"thread" CUDA builtins of oracle/ref_shim/cuda_runtime.h
#include "cuda_runtime.h"
thread_local uint3 blockIdx, threadIdx;
thread_local dim3 blockDim, gridDim;
