// Protocol experiments for a substep scheduler: per-XCD ticket counters, per-item `done` counters, payload hand-over
// between waves.  Every loop is iteration-capped: the program cannot hang.  Variants: 0 = tickets only; 1 = + done hand-shake
// (agent-scope load / store); 2 = hand-shake through atomic RMW (atomicAdd(p, 0) / atomicExch); 3 = variant 2 + payload check.
// hipcc --offload-arch=gfx950 -O3 -o ticket_proto ticket_proto.hip && ./ticket_proto [n_items] [rounds] [nq]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__device__ __forceinline__ int xcc_id() { return (int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15u); }
__global__ void k(int variant, int* tick, int* done, double* payload, int* stats, int* seen, int n, int rounds, int nq) {
  const int lane = threadIdx.x, xcc = xcc_id() % nq;
  if (lane == 0) atomicAdd(stats + xcc, 1);                       // waves per XCD
  const int cnt = (n - xcc + nq - 1)/nq, total = cnt*rounds;
  for (int guard = 0; guard < 200000; guard++) {
    int t = 0;
    if (lane == 0) t = atomicAdd(tick + 16*xcc, 1);
    t = __builtin_amdgcn_readfirstlane(__shfl(t, 0, 64));
    if (t >= total || cnt <= 0) return;
    const int round = t / cnt, item = (t % cnt)*nq + xcc;
    if (lane == 0) atomicAdd(seen + item, 1);
    if (variant == 0) continue;
    int d = 0, spins = 0;
    for (; spins < 20000; spins++) {
      if (lane == 0) d = (variant == 1) ? __hip_atomic_load(done + item, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : atomicAdd(done + item, 0);
      d = __builtin_amdgcn_readfirstlane(__shfl(d, 0, 64));
      if (d >= round) break;
      __builtin_amdgcn_s_sleep(32);
    }
    if (lane == 0) { if (spins >= 20000) atomicAdd(stats + 16, 1); if (d > round) atomicAdd(stats + 17, 1); atomicMax(stats + 19, spins); }
    if (variant == 3) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      double v = payload[(size_t)item*64 + lane];
      if (v != (double)round && lane == 0) atomicAdd(stats + 18, 1);
      payload[(size_t)item*64 + lane] = (double)(round + 1);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    if (lane == 0) { if (variant == 1) __hip_atomic_store(done + item, round + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else atomicExch(done + item, round + 1); }
  }
  if (lane == 0) atomicAdd(stats + 20, 1);                        // guard exhausted
}
int main(int argc, char** argv) {
  int n = argc > 1 ? atoi(argv[1]) : 4096, rounds = argc > 2 ? atoi(argv[2]) : 10, nq = argc > 3 ? atoi(argv[3]) : 8;
  int *tick, *done, *stats, *seen; double* payload;
  hipMalloc(&tick, 1024); hipMalloc(&done, n*4); hipMalloc(&stats, 128); hipMalloc(&seen, n*4); hipMalloc(&payload, (size_t)n*64*8);
  for (int variant = 0; variant < 4; variant++) {
    hipMemset(tick, 0, 1024); hipMemset(done, 0, n*4); hipMemset(stats, 0, 128); hipMemset(seen, 0, n*4); hipMemset(payload, 0, (size_t)n*64*8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(n), dim3(64), 0, 0, variant, tick, done, payload, stats, seen, n, rounds, nq);
    hipEventRecord(e1);
    hipError_t err = hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<int> h(32), hs(n), ht(256); hipMemcpy(h.data(), stats, 128, hipMemcpyDeviceToHost); hipMemcpy(hs.data(), seen, n*4, hipMemcpyDeviceToHost);
    hipMemcpy(ht.data(), tick, 1024, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < n; i++) if (hs[i] != rounds) bad++;
    printf("variant %d: err %d  %.2f ms  waves/xcc:", variant, (int)err, ms); for (int x = 0; x < 8; x++) printf(" %d", h[x]);
    printf("  tick:"); for (int x = 0; x < 8; x++) printf(" %d", ht[16*x]);
    printf("  items with wrong ticket count %d  spin caps %d  ahead %d  payload mismatches %d  max spins %d  guard %d\n", bad, h[16], h[17], h[18], h[19], h[20]);
    fflush(stdout);
  }
  return 0;
}
