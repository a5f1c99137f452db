// stand-in for CUDA's math_constants.h (TEST INFRASTRUCTURE ONLY): the constants the reference's kernels use
#pragma once
#define CUDART_PI_F 3.141592654f
#define CUDART_PI 3.1415926535897931e+0
#define CUDART_PIO2_F 1.570796327f
#define CUDART_PIO2 1.5707963267948966e+0
#define CUDART_PIO4_F 0.785398163f
#define CUDART_INF_F __builtin_huge_valf()
