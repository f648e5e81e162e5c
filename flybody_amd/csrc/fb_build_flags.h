// Compiler flags of every device build of the engine besides -O3 -std=c++17 (read by __graft_entry__.py and tools/build_*.sh; this file is
// part of the source hash the libraries embed, so changing a flag rebuilds them).
//
// -amdgpu-sched-strategy=max-ilp: the step kernel runs at a FIXED occupancy (launch bounds: 2 / 3 / 4 waves per SIMD) and is bound by
// dependent latency, so the machine scheduler should order for instruction-level parallelism instead of trading it for registers it
// cannot turn into more waves.  Same instructions, different order (the FP instruction counts of the two schedules are identical:
// results bit-equal); 1 400 fewer hazard s_nop in the FP64 kernel; +0.6 % env-steps/s.
//
// FB_HIPCC_FLAGS: -mllvm -amdgpu-sched-strategy=max-ilp
