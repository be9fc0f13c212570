/* stand-in for <hip/hip_runtime.h> when minppo_amd/csrc/model_view.h (the blob layout) is included by the CPU twin: g++, no HIP */
#pragma once
#define __host__
#define __device__
