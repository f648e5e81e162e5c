// Latency microbenchmarks on one wavefront (gfx950): what does the serial chain of the PGS block update actually pay for?
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 4096
__device__ __forceinline__ double rdl(double v, int s) { int lo = __builtin_amdgcn_readlane(__double2loint(v), s), hi = __builtin_amdgcn_readlane(__double2hiint(v), s); return __hiloint2double(hi, lo); }
__global__ void k(double* out, long long* cyc, double a, double b, int mode) {
  __shared__ double lds[256];
  lds[threadIdx.x] = a + threadIdx.x; __syncthreads();
  double x = a + threadIdx.x*1e-9, y = b, z = a*0.5, w = b*0.25;
  float xf = (float)x, yf = (float)y;
  long long t0 = clock64();
  if (mode == 0) { for (int i = 0; i < N; i++) x = __builtin_fma(x, y, y); }                       // dependent f64 fma
  if (mode == 1) { for (int i = 0; i < N; i += 4) { x = __builtin_fma(x, y, y); z = __builtin_fma(z, y, y); w = __builtin_fma(w, y, y); a = __builtin_fma(a, y, y); } x += z + w + a; }   // 4 independent chains
  if (mode == 2) { for (int i = 0; i < N; i++) xf = __builtin_fmaf(xf, yf, yf); x = xf; }          // dependent f32 fma
  if (mode == 3) { for (int i = 0; i < N; i++) x = 1.0/x + y; }                                     // IEEE f64 division + add
  if (mode == 4) { for (int i = 0; i < N; i++) { double r = __builtin_amdgcn_rcp(x); r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r); r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r); x = r + y; } }   // rcp + 2 Newton + add
  if (mode == 5) { for (int i = 0; i < N; i++) x = rdl(x, (i & 31)) + y; }                          // readlane (uniform index) + add
  if (mode == 6) { for (int i = 0; i < N; i++) { if (x > 1e300) x = x*y; else x = x + y; if (__builtin_amdgcn_readfirstlane(__double2hiint(x)) == 12345) break; } }   // compare + uniform branch per iteration
  if (mode == 7) { for (int i = 0; i < N; i++) x = lds[(__double2loint(x) & 63)] + y; }              // dependent LDS read + add
  if (mode == 8) { for (int i = 0; i < N; i++) x = sqrt(x) + y; }                                    // IEEE f64 sqrt + add
  if (mode == 9) { for (int i = 0; i < N; i++) { double r = __builtin_amdgcn_rsq(x); r = r*__builtin_fma(-0.5*x*r, r, 1.5); x = r + y; } }   // rsq + 1 Newton + add
  if (mode == 10) { for (int i = 0; i < N; i++) x = (x > y) ? x*0.999 : x + y; }                     // compare + select (cndmask) + ops
  if (mode == 11) { for (int i = 0; i < N; i++) x = __builtin_fma(x, y, y); }                       // same as 0, run with 2/4 waves per SIMD by the host
  long long t1 = clock64();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x*64 + threadIdx.x] = x;
}
int main() {
  double* out; long long* cyc; hipMalloc(&out, 8*64*8192 + 4096); hipMalloc(&cyc, 8*8192 + 64);
  const char* names[] = {"dependent fma f64", "4 independent fma f64 chains (per op)", "dependent fma f32", "IEEE div f64 + add", "rcp+2NR f64 + add", "readlane f64 + add", "cmp + uniform branch + add",
                         "dependent LDS read + add", "IEEE sqrt f64 + add", "rsq+1NR f64 + add", "cmp+select f64 (2 ops)", "dependent fma f64"};
  for (int mode = 0; mode <= 10; mode++) {
    k<<<1, 64>>>(out, cyc, 1.000001, 0.999999, mode); hipDeviceSynchronize();
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-42s %7.1f cycles per iteration (1 wave on the device)\n", names[mode], (double)c/N);
  }
  // SIMD sharing: the same dependent chain with 1, 2, 4, 8 waves per SIMD (1024 SIMDs)
  for (int wps : {1, 2, 4, 8}) {
    int blocks = 1024*wps;
    k<<<blocks, 64>>>(out, cyc, 1.000001, 0.999999, 0); hipDeviceSynchronize();
    long long c[64]; hipMemcpy(c, cyc, 8*64, hipMemcpyDeviceToHost);
    printf("dependent fma f64 with %d waves per SIMD: %7.1f cycles per iteration per wave\n", wps, (double)c[0]/N);
  }
  return 0;
}
