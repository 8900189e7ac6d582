// Launchers of the LDS-resident PEG kernels (peg_lds.hip), called from the C ABI in conv.hip.  Return 0 when launched, 1 when the
// geometry is not covered (the caller then runs the first-generation kernels).
#pragma once
#include "common.h"

bool peg_lds_supported(int64_t B, int D1, int D2, int D3, int C, int dtype);
// dir = +1: y = x + bias + conv(x) ; dir = -1: y = x + conv^T(x)
// rres (forward only, may be null): the compensated residual stream -- s = x + ein (may be null) + conv(x), y = bf16(s), rres = bf16(s - y)
int peg_lds_march(const void* x, const float* w, const float* bias, void* y, int64_t B, int D1, int D2, int D3, int C, int dir, hipStream_t s,
                  const void* ein = nullptr, void* rres = nullptr);
// partial weight gradients part[groups][C][28] (27 taps + bias); *groups <= peg_lds_wgrad_groups(B, D2, C)
int64_t peg_lds_wgrad_groups(int64_t B, int D2, int C);
int peg_lds_wgrad(const void* dy, const void* x, float* part, int64_t B, int D1, int D2, int D3, int C, int* groups, hipStream_t s);
