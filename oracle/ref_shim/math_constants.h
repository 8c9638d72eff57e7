// TEST INFRASTRUCTURE: CUDA's math_constants.h, the one constant the reference uses.
#ifndef SDN_REF_SHIM_MATH_CONSTANTS_H
#define SDN_REF_SHIM_MATH_CONSTANTS_H
#define CUDART_PI_F 3.141592654f
#endif
