// Cross-lane helpers shared by the device sources (internal): lane-half / lane-row swaps of doubles, DPP row rotations,
// and the wave-wide reduce-scatter of four values built from them.
#pragma once
#include <hip/hip_runtime.h>

namespace cmx {

typedef int cmx_i4 __attribute__((ext_vector_type(4)));
typedef int cmx_i16v __attribute__((ext_vector_type(16)));

// v_permlane32_swap exchanges the upper half of its first operand with the lower half of its second one,
// v_permlane16_swap does the same for the odd / even rows of 16 lanes: after swap(a, b) the sum a + b holds, in the
// lanes that keep a's group, own + partner's a, and in the others own + partner's b -- no LDS round trip.
__device__ __forceinline__ void swap32(double& a, double& b) {
  const unsigned long long ua = __builtin_bit_cast(unsigned long long, a), ub = __builtin_bit_cast(unsigned long long, b);
  const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)ua, (unsigned)ub, false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(ua >> 32), (unsigned)(ub >> 32), false, false);
  a = __builtin_bit_cast(double, ((unsigned long long)hi[0] << 32) | lo[0]);
  b = __builtin_bit_cast(double, ((unsigned long long)hi[1] << 32) | lo[1]);
}
__device__ __forceinline__ void swap16(double& a, double& b) {
  const unsigned long long ua = __builtin_bit_cast(unsigned long long, a), ub = __builtin_bit_cast(unsigned long long, b);
  const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)ua, (unsigned)ub, false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(ua >> 32), (unsigned)(ub >> 32), false, false);
  a = __builtin_bit_cast(double, ((unsigned long long)hi[0] << 32) | lo[0]);
  b = __builtin_bit_cast(double, ((unsigned long long)hi[1] << 32) | lo[1]);
}
template <int CTRL>
__device__ __forceinline__ double mica_dpp_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// totals over the wave of four values at once: lanes with lane >> 4 == r end with the total of value r
__device__ __forceinline__ double mica_reduce4(double p0, double p1, double p2, double p3) {
  // totals over the wave of four values at once: lanes with lane >> 4 == r end with the total of value r
  swap32(p0, p2);
  swap32(p1, p3);
  double k0 = p0 + p2, k1 = p1 + p3;
  swap16(k0, k1);
  double s = k0 + k1;
  s += mica_dpp_f64<0x128>(s);   // row_ror:8
  s += mica_dpp_f64<0x124>(s);   // row_ror:4
  s += mica_dpp_f64<0x122>(s);   // row_ror:2
  s += mica_dpp_f64<0x121>(s);   // row_ror:1
  return s;
}

}  // namespace cmx
