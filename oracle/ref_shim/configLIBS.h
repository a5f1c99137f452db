// what CMake would generate from configLIBS.h.in for an OpenMP + OpenCV3 build without CUDA/Ceres
#ifndef I3D_LINE3D_PP_LIBS_CONFIG_H_
#define I3D_LINE3D_PP_LIBS_CONFIG_H_
#define L3DPP_OPENMP 1
#define L3DPP_OPENCV3 1
#endif
