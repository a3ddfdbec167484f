// Stand-in for <hip/hip_runtime.h> so that the device headers of winterfell_amd/csrc compile as plain host C++
// (tests/cpp/ntt_big_host_test.cpp: the three-step NTT pass emulated lane by lane on the CPU).
#pragma once
#define __device__
#define __host__
#define __global__
#define __shared__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
