// LDS addressing and wave-level reductions shared by the on-chip kernels (vsm_fused.hip, vsm_strip.hip).
#pragma once
#include "vsm_common.h"

namespace vsm {

// Column-major NP x NP image of an operator in LDS (NP multiple of 32).  The row index is XOR-swizzled
// by the column so that the MFMA operand fetch patterns (A: 16 rows x {k..k+3} columns; B / accumulator:
// {k..k+3} rows x 16 columns) hit distinct banks for ds_read_b64 / ds_read_b32.
template <int NP>
__device__ __forceinline__ int lidx(int a, int b) {
  return (a ^ (((b & 1) << 4) | (((b >> 1) & 7) << 1))) + NP * b;
}

// ---- deterministic wave sum (DPP row rotations + 4 readlanes) ----------------------------------------
__device__ __forceinline__ float dpp_ror_add(float x, const int ctrl_is_8_4_2_1) {
  float y;
  switch (ctrl_is_8_4_2_1) {
    case 8: y = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x128, 0xf, 0xf, false)); break;
    case 4: y = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x124, 0xf, 0xf, false)); break;
    case 2: y = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x122, 0xf, 0xf, false)); break;
    default: y = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x121, 0xf, 0xf, false)); break;
  }
  return x + y;
}
__device__ __forceinline__ float wave_sum(float x) {
  x = dpp_ror_add(x, 8);
  x = dpp_ror_add(x, 4);
  x = dpp_ror_add(x, 2);
  x = dpp_ror_add(x, 1);  // every lane of a 16-lane row now holds the row sum
  const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 0));
  const float b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 16));
  const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 32));
  const float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 48));
  return (a + b) + (c + d);
}
__device__ __forceinline__ float to_float_up(double x) { return __double2float_ru(x); }
__device__ __forceinline__ float to_float_up(float x) { return x; }

}  // namespace vsm
