// Evidence for cfr_rows_kernel.hip::norm_row: (x + eps) / s computed as hipcc's own f64 division sequence minus
// v_div_scale / v_div_fixup, with the reciprocal refinement shared per denominator, against the plain `/` operator, on
// operands from the kernel's range (1e-80 <= a <= s <= H + 1, including the extremes).  Prints the number of mismatches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ uint64_t rng(uint64_t& s) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
__device__ double u01(uint64_t& s) { return (rng(s) >> 11) * (1.0 / 9007199254740992.0); }
__global__ void k(unsigned long long* mism, unsigned long long* total, int iters) {
  uint64_t st = 0x9E3779B97F4A7C15ull * (blockIdx.x * blockDim.x + threadIdx.x + 1);
  unsigned long long bad = 0, n = 0;
  for (int it = 0; it < iters; ++it) {
    double r[6], ssum = 0;
    const int mode = rng(st) & 7;
    for (int h = 0; h < 6; ++h) {
      double v = u01(st);
      if (mode == 1) v *= 1e-30 * u01(st);
      if (mode == 2) v = (rng(st) & 1) ? 0.0 : v;
      if (mode == 3) v = ldexp(v, -(int)(rng(st) % 260));
      if (mode == 4) v = (h == 0) ? v : 0.0;
      if (mode == 5) v = 1e-80 * u01(st);
      r[h] = v;
    }
    for (int h = 0; h < 6; ++h) ssum += r[h] + 1e-80;
    double y = __builtin_amdgcn_rcp(ssum);
    double e = __builtin_fma(-ssum, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-ssum, y, 1.0);
    y = __builtin_fma(y, e, y);
    for (int h = 0; h < 6; ++h) {
      const double num = r[h] + 1e-80;
      const double q0 = num * y;
      const double rem = __builtin_fma(-ssum, q0, num);
      const double q = __builtin_fma(rem, y, q0);
      const double ref = num / ssum;
      bad += (__double_as_longlong(q) != __double_as_longlong(ref));
      ++n;
    }
  }
  atomicAdd(mism, bad);
  atomicAdd(total, n);
}
int main() {
  unsigned long long *d, h[2] = {0, 0};
  (void)hipMalloc(&d, 16);
  (void)hipMemcpy(d, h, 16, hipMemcpyHostToDevice);
  k<<<1024, 256>>>(d, d + 1, 2000);
  (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  printf("divisions %llu, results differing from the / operator: %llu\n", h[1], h[0]);
  return h[0] != 0;
}
