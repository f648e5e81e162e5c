// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for THIS engine's access widths.
// The MI355X guide calibrates FETCH_SIZE only for 16 B/lane streams (reported = 1/2 of the bytes).  The step kernel moves
// 8 B/lane (FP64 build) and 4 B/lane (FP32 build, integer arrays) in 512 B / 256 B row segments of a 300 KB row per environment.
// Each kernel below moves a KNOWN number of bytes through HBM (buffers far larger than the 256 MiB Infinity Cache, every byte
// touched exactly once); tools/calibrate_traffic.sh runs it under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` and
// divides what the counters report by what was moved.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// coalesced stream, T per lane
template <typename T> __global__ void k_read(const T* __restrict__ p, size_t n, T* __restrict__ sink) {
  T acc = 0;
  for (size_t i = (size_t)blockIdx.x*blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x*blockDim.x) acc += p[i];
  if (acc == (T)123456789) sink[0] = acc;
}
template <typename T> __global__ void k_write(T* __restrict__ p, size_t n, T v) {
  for (size_t i = (size_t)blockIdx.x*blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x*blockDim.x) p[i] = v + (T)i;
}
// the engine's pattern: one wave per "environment row"; the wave reads / writes segments of 64 consecutive doubles at scattered
// offsets inside its own row (rows are `stride` doubles apart), every segment exactly once
__global__ void k_rows_read(const double* __restrict__ p, size_t stride, int nseg, double* __restrict__ sink) {
  const double* row = p + (size_t)blockIdx.x*stride;
  double acc = 0;
  for (int s = 0; s < nseg; s++) acc += row[(size_t)((s*37) % nseg)*64 + threadIdx.x];       // (a permutation of the segments: 37 is coprime to nseg)
  if (acc == 123456789.0) sink[0] = acc;
}
__global__ void k_rows_write(double* __restrict__ p, size_t stride, int nseg, double v) {
  double* row = p + (size_t)blockIdx.x*stride;
  for (int s = 0; s < nseg; s++) row[(size_t)((s*37) % nseg)*64 + threadIdx.x] = v + s;
}

int main(int argc, char** argv) {
  const size_t bytes = (size_t)2 << 30;                       // 2 GiB per pass
  void* buf; CHK(hipMalloc(&buf, bytes)); CHK(hipMemset(buf, 0, bytes));
  double* sink; CHK(hipMalloc((void**)&sink, 64));
  const int mode = argc > 1 ? atoi(argv[1]) : -1;
  const int grid = 256*8*4, block = 256;
  // every mode is its own kernel name in the counter csv; one launch each
  if (mode < 0 || mode == 0) hipLaunchKernelGGL(k_read<double>, dim3(grid), dim3(block), 0, 0, (const double*)buf, bytes/8, sink);
  if (mode < 0 || mode == 1) hipLaunchKernelGGL(k_read<float>, dim3(grid), dim3(block), 0, 0, (const float*)buf, bytes/4, (float*)sink);
  if (mode < 0 || mode == 2) hipLaunchKernelGGL(k_write<double>, dim3(grid), dim3(block), 0, 0, (double*)buf, bytes/8, 1.0);
  if (mode < 0 || mode == 3) hipLaunchKernelGGL(k_write<float>, dim3(grid), dim3(block), 0, 0, (float*)buf, bytes/4, 1.0f);
  // 4096 rows of 64 Ki doubles (512 KiB) = 2 GiB; 1021 segments of 512 B per row are touched (1021 is prime: s*37 mod 1021 is a permutation)
  const size_t stride = 65536; const int nseg = 1021;
  if (mode < 0 || mode == 4) hipLaunchKernelGGL(k_rows_read, dim3(4096), dim3(64), 0, 0, (const double*)buf, stride, nseg, sink);
  if (mode < 0 || mode == 5) hipLaunchKernelGGL(k_rows_write, dim3(4096), dim3(64), 0, 0, (double*)buf, stride, nseg, 2.0);
  CHK(hipDeviceSynchronize());
  printf("{\"k_read<double>\": %zu, \"k_read<float>\": %zu, \"k_write<double>\": %zu, \"k_write<float>\": %zu, \"k_rows_read\": %zu, \"k_rows_write\": %zu}\n",
         bytes, bytes, bytes, bytes, (size_t)4096*nseg*512, (size_t)4096*nseg*512);
  return 0;
}
