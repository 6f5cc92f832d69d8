// Entry points of the convolution kernel units, called by the dispatcher (conv.hip: demfi_conv2d).  Host-side C++, not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include "demfi_hip.h"

// conv_general.hip: every shape (fp16 / fp32), one workgroup per 8 x 32 tile
int demfi_conv_general_launch(const demfi_conv* h, const demfi_conv* dev, hipStream_t st, size_t lds);
// conv_c64.hip: 3x3, one 64-channel NHWC piece, 32 / 64 couts (staged-store kernel for 64); *fall_through: an ablation build asked for the general kernel
int demfi_c64_launch(const demfi_conv* h, const demfi_conv* dev, hipStream_t st, bool* fall_through);
// conv_narrow.hip: 3x3 / 7x7 over one chunk of 16 / 32 / 64 channels; thin: planar fp32 outputs.  *handled = false: no instantiation for the shape
int demfi_narrow_launch(const demfi_conv* h, const demfi_conv* dev, hipStream_t st, bool thin, bool* handled);
// conv_sep.hip: the round-1..5 SepConvGRU kernel (1x5 / 5x1, two 64-channel pieces)
bool demfi_sep_eligible(const demfi_conv* h);
int demfi_sep_launch(const demfi_conv* h, const demfi_conv* dev, hipStream_t st, bool* fall_through);
// conv_wstream.hip: Ch_Reducer's streamed-weight kernel (ks 7, nch 2) and its 3x3 / 32-cout instantiation (ks 3, nch 1)
bool demfi_wstream_eligible(const demfi_conv* h, int ks = 7, int nch = 2);
bool demfi_wstream3_on();
int demfi_wstream_launch(const demfi_conv* h, const demfi_conv* dev, hipStream_t st);
int demfi_wstream3_launch(const demfi_conv* h, const demfi_conv* dev, hipStream_t st);
// experiment builds (no-ops in the product): per-unit knob word and phase-trace buffer
void demfi_c64_set_knob(int k);
void demfi_narrow_set_knob(int k);
int demfi_c64_trace_collect(unsigned long long* acc);
int demfi_narrow_trace_collect(unsigned long long* acc);
int demfi_sep_trace_collect(unsigned long long* acc);
int demfi_wstream_trace_collect(unsigned long long* acc);
