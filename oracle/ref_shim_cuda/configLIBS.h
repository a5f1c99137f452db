// what CMake would generate from configLIBS.h.in for an OpenMP + OpenCV3 build WITH CUDA (no Ceres): the second build
// of the reference under oracle/_ref (libl3d_ref_cuda.so), in which the reference's CUDA path runs as host code
// through oracle/ref_shim_cuda/cuda_runtime.h -- test infrastructure, used to pin the replicator-dynamics diffusion
// (performRDD / cudawrapper.cu), the one step of the reference that has no CPU path.
#ifndef I3D_LINE3D_PP_LIBS_CONFIG_H_
#define I3D_LINE3D_PP_LIBS_CONFIG_H_
#define L3DPP_OPENMP 1
#define L3DPP_OPENCV3 1
#define L3DPP_CUDA 1
#endif
