// Compiler flags of every device build of the engine besides -O3 -std=c++17 (read by __graft_entry__.py and tools/build_*.sh; this file is
// part of the source hash the libraries embed, so changing a flag rebuilds them).
//
// -amdgpu-sched-strategy=max-ilp: the step kernel runs at a FIXED occupancy (launch bounds: 2 / 3 / 4 waves per SIMD) and is bound by
// dependent latency, so the machine scheduler should order for instruction-level parallelism instead of trading it for registers it
// cannot turn into more waves.  Same instructions, different order (the FP instruction counts of the two schedules are identical:
// results bit-equal); 1 400 fewer hazard s_nop in the FP64 kernel; +0.6 % env-steps/s.
//
// -fapprox-func (round 6): the `afn` fast-math flag ALONE (no reassociation, no contraction change, no finite-math assumptions).  What it changes
// in this code: the remaining plain `a / b` of the lane-parallel stages (ray casts of the touch sensors, closed-form collision pairs, impedances,
// the noslip set-up) lower to v_rcp_f64 + two Newton steps + one residual correction (<= 1 ulp) instead of the IEEE sequence with v_div_scale /
// v_div_fmas / v_div_fixup (~25 dependent instructions): 1-3 % fewer instructions in seven stage functions, +0.3 % env-steps/s, divergence from
// the oracle over 100 control steps unchanged to the digit (profiles/r6/ab_reciprocals.txt).
//
// FB_HIPCC_FLAGS: -mllvm -amdgpu-sched-strategy=max-ilp -fapprox-func
