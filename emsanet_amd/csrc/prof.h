// Per-launch HIP-event timing of the convolution kernels (bench.py's roofline numbers).
// Defined in conv_mfma.hip; events are recorded on the launch stream around every n-th launch of
// a kernel class and read back through the emsa_prof_* C-ABI after the stream is synchronised.
#pragma once
#include <hip/hip_runtime.h>

constexpr int kProfClasses = 14;
constexpr int kProfClassWino = 8;      // Winograd F(2,3) forward / data gradient of the 1-D (3x1 / 1x3) convs
constexpr int kProfClassConvH = 9;     // 16-bit implicit GEMM (conv_h.hip)
constexpr int kProfClassWgradH = 10;   // weight gradients on 16-bit activations
// SURVEY.md 0.3 / VERDICT r2: the 1-D NBt1D kernel and the dense 3x3 kernel are reported apart
constexpr int kProfClassWino3x3 = 11;  // the same kernel as row-Winograd on the dense 3x3 convs
constexpr int kProfClassWgrad3x3 = 12; // Winograd F(3,2) weight gradient of the dense 3x3 convs
constexpr int kProfClassConvRS = 13;   // register-stationary streaming kernel of the 16-bit 1-D convs (conv_rs.hip)

// returns a slot id (or -1 when this launch is not sampled); `flops` = algorithmic FLOPs, `bytes` =
// algorithmic HBM bytes (every tensor the launch reads or writes counted once; 0 = not tracked)
int emsa_prof_begin(int cls, double flops, hipStream_t st, double bytes = 0.0);
void emsa_prof_end(int slot, hipStream_t st);
