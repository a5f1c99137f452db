// stand-in for <cuda.h>: everything lives in cuda_runtime.h (oracle/ref_shim_cuda, TEST INFRASTRUCTURE ONLY)
#pragma once
#include "cuda_runtime.h"
