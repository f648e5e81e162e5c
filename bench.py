#!/usr/bin/env python3
"""Headline benchmark: env steps/sec of batched walk_imitation random-action rollouts.

`python bench.py --gpus N --steps K --warmup W`: one rank per GPU.  Under torch.distributed.run (the
driver's N > 1 launch, WORLD_SIZE set) the process is one rank of the job; started bare with
`--gpus N > 1` it re-executes itself under torch.distributed.run with N ranks (RCCL).  A "step" is
one control step (10 physics substeps + observation / reward / termination epilogue) of every
environment of the batch; environments are sharded across ranks with no data-path collective ("weak" scaling: 4096 envs per GPU, BASELINE.json configs[1]).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# words read + written per env control step (qpos 109 + qvel 108 + act 59 in and out; action 59 in;
# obs 741 + reward/discount/step_type 3 out).  State words are 4 B (f32 build) or 8 B (f64 build), action/obs/reward are f32.
ALGO_BYTES_PER_ENV_STEP = {32: 5420.0, 64: 2*276*8.0 + (59 + 741 + 3)*4.0}
# instrumented count of the FP64 CPU oracle (tools/flopcount/count_flops.py, profiles/r1/oracle_flop_count.jsonl): adds,
# multiplies, divisions and square roots of one walk_imitation control step, mean over 200 steps in contact under the
# bench's action distribution; replaces SURVEY.md 8(d)'s provisional 9 MFLOP.  Counted on the algorithm the oracle runs NOW
ALGO_FLOP_PER_ENV_STEP = 3.14e6                # round 3 (Newton solver): 3 138 048; rounds 1-2 (PGS): 3 443 346 (profiles/r3/oracle_flop_count.jsonl)
# flight_imitation (configs[3]): SURVEY.md 8(d) -- 1 156 B and ~0.53 MFLOP per env control step (4 substeps of 5e-5 s, nv 42, no floor)
FLIGHT_BYTES_PER_ENV_STEP = {32: 1156.0, 64: 2*(43 + 42)*8.0 + (12 + 104 + 3)*4.0}
FLIGHT_FLOP_PER_ENV_STEP = 0.534e6
HBM_PEAK_GBS = 8000.0                  # MI355X_MICROARCH.md
VALU_PEAK_TFLOPS = {32: 157.3, 64: 78.6}


def cpu_baseline(seconds=12.0):
    """Own FP64 CPU oracle (a 'port', NOT CPU MuJoCo) timed on this box's host cores.

    One environment on one thread first, then every usable core: two environments per thread, each running its own action
    sequence for a block of control steps with NO barrier between steps (oracle/fbo_env.c: fbo_env_rollout_batch), threads
    pinned to cores.  Reports the aggregate, the thread count and the parallel efficiency against the single-core rate."""
    import numpy as np
    os.environ.setdefault('OMP_PROC_BIND', 'spread'); os.environ.setdefault('OMP_PLACES', 'cores')     # before libgomp starts
    from flybody_amd.model_blob import load_npz, pack_model
    from flybody_amd.reference import default_walking_reference
    from oracle import fbo
    arrays = load_npz(os.path.join(ROOT, 'flybody_amd', 'assets', 'walk_imitation.npz'))
    om = fbo.OracleModel(pack_model(arrays))
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    # the container's CPU bandwidth limit (cgroup v2 cpu.max = "<quota> <period>"): the GPU boxes of this pool show 256 logical
    # CPUs but a quota of 16 -- more runnable threads than the quota only get throttled (measured: 32 threads burst to 40 k
    # env-steps/s for a fraction of a second, 64+ threads sustain 12-20 k), so the baseline runs one thread per granted CPU
    quota = None
    try:
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()
        if q != 'max':
            quota = max(1, int(round(int(q)/int(per))))
    except (OSError, ValueError):
        pass
    budget_cores = min(usable, quota) if quota else usable
    qp, qv = default_walking_reference()
    rng = np.random.default_rng(0)

    def make(n):
        envs = []
        for _ in range(n):
            d = fbo.OracleData(om); d.configure_env(qp, qv, terminal_com_dist=float('inf')); d.env_reset(); envs.append(d)
        return envs

    def run(envs, nthreads, block, budget):
        n = 0; t0 = time.perf_counter()
        while True:
            fbo.rollout_batch(envs, np.clip(rng.normal(size=(len(envs), block, 59)), -1, 1), nthreads)
            n += len(envs)*block
            if time.perf_counter() - t0 >= budget:
                break
        return n/(time.perf_counter() - t0), n

    e1 = make(1)
    run(e1, 1, 20, 0.2)
    r1, n1 = run(e1, 1, 50, seconds*0.25)
    best = (r1, 1, n1)
    for nt in sorted({budget_cores}):
        if nt <= 1:
            continue
        envs = make(2*nt)
        run(envs, nt, 10, 1.0)                                  # warm-up: thread team, caches, and the cgroup's burst allowance
        r, n = run(envs, nt, 40, seconds*0.6)
        if r > best[0]:
            best = (r, nt, n)
    return {'value': best[0], 'unit': 'env steps/sec', 'cores': best[1], 'kind': 'port',
            'single_core_value': r1, 'usable_cores': usable, 'cgroup_cpu_quota': quota, 'parallel_efficiency': best[0]/(best[1]*r1),
            'sample': f'{best[2]} walk_imitation control steps (two envs per thread, 40-step blocks without a per-step barrier, N(0,1) '
                      f'actions clipped to [-1,1]) on the own FP64 C oracle (gcc -O3 -march=native, OpenMP, {best[1]} pinned threads) '
                      f'-- NOT CPU MuJoCo'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--envs-per-gpu', type=int, default=4096)
    ap.add_argument('--preroll', type=int, default=235,
                    help='untimed control steps before the warm-up.  They STAGGER the episode phases: the environments whose global id is k modulo '
                         'the pre-roll length are reset again before pre-roll step k, so that after one episode length (235 control steps, '
                         'walk_imitation.py:104-105) the phases are spread evenly -- the steady state of a long-running actor pool, where every '
                         'launch carries n_env/236 auto-resetting environments -- instead of all 4096 resetting in ONE launch of the timed window')
    ap.add_argument('--seed', type=int, default=1234, help='key of the per-environment Philox action streams (fb_random_actions)')
    # The headline leg is the FP64 build: it reproduces the FP64 CPU oracle step for step (<= 2e-11 relative over 100 control
    # steps), i.e. it is inside north_star's 1e-4 tolerance for every environment.  The FP32 build (2.3x faster) drifts
    # chaotically like any single-precision MuJoCo (median environment inside 1e-4 after 100 physics steps, not every one);
    # its throughput is reported next to the headline as `f32_mode`, never as `value`.
    ap.add_argument('--precision', type=int, default=int(os.environ.get('FB_PRECISION', '64')), choices=[32, 64])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-f32-leg', action='store_true', help='skip the secondary FP32-build measurement')
    ap.add_argument('--no-split-leg', action='store_true', help='skip the secondary two-half-batches-on-two-streams measurement')
    ap.add_argument('--no-secondary-configs', action='store_true', help='skip the flight_imitation (configs[3]) and DMPO (configs[2]) legs')
    ap.add_argument('--no-parity-sample', action='store_true', help='skip the post-run replay of sampled environments on the CPU oracle')
    ap.add_argument('--no-flight-leg', action='store_true', help='skip the flight_imitation leg (configs[3]) only')
    ap.add_argument('--dmpo-envs', type=int, default=4096, help='environments per GPU of the DMPO leg (configs[2] on one rank, configs[4] on several)')
    # DMPO leg at the REFERENCE's parameters (train_dmpo_ray.py:105-137, ray_distributed_dmpo.py:82-96): replay table of 4 000 000 items,
    # min_replay_size 10 000, 15 samples per insert, batch 256, 20 sampled actions; >= 100 timed control steps
    ap.add_argument('--dmpo-iters', type=int, default=100); ap.add_argument('--dmpo-warmup', type=int, default=6); ap.add_argument('--dmpo-min-replay', type=int, default=10_000)
    ap.add_argument('--dmpo-replay-capacity', type=int, default=4_000_000)
    ap.add_argument('--dmpo-timeout', type=float, default=240.0, help='seconds after which a DMPO job is killed (its leg is lost, the benchmark line is not)')
    ap.add_argument('--no-dmpo-f32', action='store_true', help='skip the FP32-physics repeat of the DMPO leg (reported beside the FP64 figure on one rank)')
    ap.add_argument('--rccl-dry-run', action='store_true', help='with --gpus N > 1: use RCCL when N devices are visible; otherwise run the N ranks on the '
                    'one visible device over gloo (same code path up to the backend) and say `rccl: unexercised` in the JSON')
    args = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # bare `python bench.py --gpus N`: become the launcher of N ranks (one per GPU, RCCL), same arguments
        import socket
        import subprocess
        import torch
        dry_env = {}
        if 'FB_BENCH_DEVICE' not in os.environ and torch.cuda.device_count() < args.gpus:
            if not args.rccl_dry_run:
                raise SystemExit(f'bench.py: --gpus {args.gpus} but only {torch.cuda.device_count()} GPU(s) are visible (--rccl-dry-run: run the ranks on one device over gloo)')
            dry_env = {'FB_BENCH_DEVICE': '0', 'FB_BENCH_BACKEND': 'gloo'}
        with socket.socket() as s:
            s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]
        env = dict(os.environ, MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'), **dry_env)
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
               '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))

    import torch
    import numpy as np
    from flybody_amd import engine
    from flybody_amd.reference import default_walking_reference
    from flybody_amd.sharding import staggered_preroll

    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)')
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the engine has no CPU fallback)')
    # functional testing of the N > 1 path on a one-GPU box: FB_BENCH_DEVICE pins every rank to one device and
    # FB_BENCH_BACKEND=gloo replaces RCCL (two ranks cannot share a device under RCCL); never set by the driver
    backend = os.environ.get('FB_BENCH_BACKEND', 'nccl')
    if 'FB_BENCH_DEVICE' in os.environ:
        local_rank = int(os.environ['FB_BENCH_DEVICE'])
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend)
        # the collective path is CHECKED before it is reported: a SUM all-reduce of the rank ids on the device (N (N - 1) / 2 on every
        # rank) and an all-gather that must return 0 .. N-1 in order -- `rccl: exercised` below is printed only behind this
        dev_t = 'cuda' if backend == 'nccl' else 'cpu'
        t = torch.tensor([float(rank)], device=dev_t, dtype=torch.float32)
        dist.all_reduce(t)
        g = [torch.zeros(1, device=dev_t, dtype=torch.float32) for _ in range(world)]
        dist.all_gather(g, torch.tensor([float(rank)], device=dev_t, dtype=torch.float32))
        if float(t.item()) != world*(world - 1)/2 or [int(x.item()) for x in g] != list(range(world)):
            raise SystemExit(f'bench.py: collective check failed on rank {rank}: all_reduce(SUM) of rank ids = {float(t.item())}, all_gather = {[int(x.item()) for x in g]}')
    n_env = args.envs_per_gpu
    model = engine.Model.from_asset('walk_imitation')
    # FP64 batches beyond the 2048 resident slots of the default build run on the 12-environments-per-CU build of the same kernel
    # (engine.HIP_LIB_DENSE, 3072 resident): with the substep scheduler the extra residency pays in lock-step too (8.2 against 9.0 ms
    # for 4096 environments; FB_BENCH_DEFAULT_BUILD=1 keeps the default build).  Same source, same arithmetic, own parity tests.
    dense_headline = (args.precision == 64 and n_env > 2048 and os.path.exists(engine.HIP_LIB_DENSE) and os.environ.get('FB_BENCH_DEFAULT_BUILD') is None)
    model_headline = engine.Model.from_asset('walk_imitation', dense=True) if dense_headline else model
    qp, qv = default_walking_reference()
    stream = torch.cuda.current_stream().cuda_stream
    nu = model.dim('nu')

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # sampled environments for the oracle replay: both ends and the middle of the shard plus a spread -- with the staggered
    # pre-roll they sit at different episode phases, some of them cross their auto-reset inside the timed region
    sample_ids = np.unique(np.concatenate([[0, 1, n_env - 2, n_env - 1, n_env//2 - 1, n_env//2], np.linspace(0, n_env - 1, 12).astype(int)]))[:16]
    id_base = rank*n_env                       # global id of this rank's environment 0: action streams and stagger groups are keyed by it
    P = max(1, args.preroll)

    def run_leg(precision, extras=None):
        """P staggering + W warm-up control steps untimed, then K timed control steps of the whole batch; returns (seconds, kernel ms
        total, launches, finite).  extras (dict): filled with the end state of the sampled environments and the FB_WARN population."""
        batch = engine.Batch(model_headline if precision == args.precision else model, n_env, device=local_rank, precision=precision)
        batch.set_reference(qp, qv, terminal_com_dist=float('inf'))
        batch.reset(stream=stream)
        batch.set('SIZE_STATS', 0)                 # (the reset's forward evaluation counted one constraint set-up per environment: substeps only from here on;
                                                   #  the auto-resets inside the run, one launch in 236 per environment, still add one each)
        action = torch.empty(n_env, nu, device='cuda', dtype=torch.float32)
        aptr = action.data_ptr()

        def one_step(t):
            # per-environment Philox stream keyed by the GLOBAL environment id: N(0,1) clipped to the canonical range
            batch.random_actions(aptr, t, seed=args.seed, env_id_base=id_base, stream=stream)
            batch.step_ptr(aptr, stream)

        staggered_preroll(batch, aptr, P, args.seed, id_base, stream)          # pre-roll step k: group (global id % P == k) is reset, then everyone steps
        for k in range(args.warmup):
            one_step(P + k)
        sc0 = batch.get('STEP_COUNT').ravel().astype(np.int64)       # episode step of every environment entering the timed region
        barrier()
        batch.timing_begin(stream)
        t0 = time.perf_counter()
        for k in range(args.steps):
            one_step(P + args.warmup + k)
        kernel_ms, nlaunch = batch.timing_end(stream)      # HIP events on the launch stream
        if extras is not None:
            extras['launch_ms'] = batch.timing_launches()  # ... and around every single launch of the timed region
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device='cuda' if backend == 'nccl' else 'cpu', dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        finite = bool(np.isfinite(batch.get('QPOS')).all())
        if extras is not None:
            sc1 = batch.get('STEP_COUNT').ravel().astype(np.int64)
            # an environment that was auto-reset inside the timed region spent ONE of its K launches on the reset (a forward pass,
            # no substeps: fb_step.hpp d_run) and shows a step count below sc0 + K
            n_reset = int((sc1 < sc0 + args.steps).sum())
            extras['auto_resets'] = {'envs_reset_inside_timed_region': n_reset,
                                     'reset_env_share': n_reset/float(n_env*args.steps),
                                     'steady_state_share': 1.0/236.0,
                                     'episode_phase_min_max_entering': [int(sc0.min()), int(sc0.max())],
                                     'staggered_preroll': P,
                                     'note': 'share of (environment, launch) pairs of the timed region that were an auto-reset instead of a control step; '
                                             'a long-running pool of 235-step episodes sits at 1/236'}
            wv = batch.get('WARN_EVER').ravel()
            extras['warn'] = {name: int(((wv & bit) != 0).sum()) for name, bit in engine.WARN_BITS.items()}
            # how close the run came to the caps and to the Newton solver's one-row-per-lane limit (FB_SIZE_STATS: over EVERY substep of
            # the pre-roll, the warm-up and the timed steps, all environments)
            ss = batch.get('SIZE_STATS').reshape(-1, 4).astype(np.int64)
            nsub_total = float(n_env)*(P + args.warmup + args.steps)*int(model.dim('nsubstep'))
            extras['sizes'] = {'max_ncon': int(ss[:, 0].max()), 'max_nefc': int(ss[:, 1].max()),
                               'substep_share_above_32_rows': float(ss[:, 2].sum()/nsub_total), 'substep_share_above_64_rows': float(ss[:, 3].sum()/nsub_total),
                               'envs_that_ever_exceeded_64_rows': int((ss[:, 3] > 0).sum()),
                               'caps': {'kernel_contacts': 64, 'kernel_rows': 192, 'mujoco_nconmax': 100, 'mujoco_njmax': 300},
                               'solver': 'Newton at every system size (one row per lane up to 64 rows: registers / LDS; beyond: d_newton_wide)',
                               'substeps_counted': nsub_total}
            extras['qpos'] = batch.get('QPOS')[sample_ids]; extras['qvel'] = batch.get('QVEL')[sample_ids]
            extras['solver_iterations_mean'] = float(batch.get('SOLVER_NITER').mean())
            extras['scheduler'] = ('substep tickets: waves draw (environment, substep) units per XCD, the batch exceeds the %d resident slots' % batch.resident_slots) if batch.substep_scheduler else 'one environment per wave, longest first'
        del batch
        return dt, kernel_ms, nlaunch, finite

    def parity_sample(extras):
        """OUTSIDE the clock: the action streams of the sampled environments (regenerated on the GPU from their global ids: the streams are
        keyed by id, not by position in the batch) are replayed on the CPU oracle from each environment's LAST explicit reset -- pre-roll
        step (global id % P) -- through its auto-reset(s); the FP64 end states are compared."""
        from flybody_amd.model_blob import pack_model
        from oracle import fbo
        total = P + args.warmup + args.steps
        tmp = engine.Batch(model, 1, device=local_rank, precision=64)          # (only its library handle is used)
        gids = torch.as_tensor((id_base + sample_ids).astype(np.int32), device='cuda')
        a = torch.empty(len(sample_ids), nu, device='cuda', dtype=torch.float32)
        rec = []
        for t in range(total):
            tmp.random_actions(a.data_ptr(), t, seed=args.seed, stream=stream, env_ids_dev_ptr=gids.data_ptr(), n=len(sample_ids))
            rec.append(a.cpu().numpy().copy())
        del tmp
        acts = np.ascontiguousarray(np.stack(rec, axis=1).astype(np.float64))            # [n_sample][steps][nu]
        om = fbo.OracleModel(pack_model(model.arrays))
        rel = lambda x, y: float(np.abs(x - y).max() / max(np.abs(y).max(), 1e-300))
        eq = ev = 0.0; nsteps = []
        for i, e in enumerate(sample_ids):
            first = int((id_base + int(e)) % P)                       # the pre-roll step before which this environment was last reset explicitly
            d = fbo.OracleData(om); d.configure_env(qp, qv, terminal_com_dist=float('inf')); d.env_reset()
            fbo.rollout_batch([d], np.ascontiguousarray(acts[i:i + 1, first:]))
            eq = max(eq, rel(extras['qpos'][i], d.field('qpos'))); ev = max(ev, rel(extras['qvel'][i], d.field('qvel')))
            nsteps.append(total - first)
        return {'n': len(sample_ids), 'control_steps_min_max': [int(min(nsteps)), int(max(nsteps))], 'env_ids': [int(e) for e in sample_ids], 'max_rel_qpos': eq, 'max_rel_qvel': ev,
                'tolerance': 1e-6, 'ok': bool(eq < 1e-6 and ev < 1e-6),
                'note': 'FP64 kernel end state of sampled environments (ids across the whole batch, different episode phases) vs the FP64 CPU oracle '
                        'replaying the same per-environment action streams from the same reset; oracle vs CPU MuJoCo stays unpinned'}

    def run_flight_leg(precision, steps, warmup):
        """BASELINE configs[3]: 8192 flight_imitation environments (WBPG + ellipsoid wing fluid forces, 4 substeps of 5e-5 s),
        U(-1, 1)^12 actions, per-environment initial wing-beat phase, auto-reset included."""
        from flybody_amd.fly_envs import flight_imitation
        n_f = 8192
        env = flight_imitation(n_env=n_f, device=local_rank, precision=precision, env_id_base=rank*n_f)
        b = env.batch
        a = torch.empty(n_f, b.model.dim('nact'), device='cuda', dtype=torch.float32)
        env.reset_all()

        def one_step(t):
            b.random_actions(a.data_ptr(), t, seed=args.seed + 1, env_id_base=rank*n_f, dist=1, stream=stream)      # U(-1, 1), keyed by global id
            b.step_ptr(a.data_ptr(), stream)
        for k in range(warmup):
            one_step(k)
        barrier()
        b.timing_begin(stream); t0 = time.perf_counter()
        for k in range(steps):
            one_step(warmup + k)
        kms, nl = b.timing_end(stream)
        barrier()
        dtf = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dtf], device='cuda' if backend == 'nccl' else 'cpu', dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX); dtf = float(t.item())
        per = (kms/1e3)/max(nl, 1)
        tfl = FLIGHT_FLOP_PER_ENV_STEP*n_f/per/1e12
        out = {'value': n_f*world*steps/dtf, 'unit': 'env steps/sec', 'envs_per_gpu': n_f, 'ms_per_step': dtf/steps*1e3, 'kernel_ms_avg': per*1e3,
               'state_finite': bool(np.isfinite(b.get('QPOS')).all()),
               'roofline': {'bound': 'valu', 'achieved': tfl, 'peak': VALU_PEAK_TFLOPS[precision], 'unit': 'TFLOP/s', 'frac': tfl/VALU_PEAK_TFLOPS[precision],
                            'algorithmic_flop_per_env_step': FLIGHT_FLOP_PER_ENV_STEP,
                            'hbm': {'achieved': FLIGHT_BYTES_PER_ENV_STEP[precision]*n_f/per/1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                    'algorithmic_bytes_per_env_step': FLIGHT_BYTES_PER_ENV_STEP[precision]}}}
        del env
        return out

    def run_dmpo_leg(n_ranks=1, precision=64):
        """BASELINE configs[2] (one rank) / configs[4] (n_ranks > 1): DMPO training at the reference's parameters -- 4096 environments
        per GPU, replay table of 4 000 000 items, min_replay_size 10 000, the reference's rate limiter (15 samples per insert = 240
        learner steps per control step of a rank's shard), >= 100 timed control steps.  Own process (it owns the torch RNG and the HIP
        graphs) -- for several ranks its own JOB: `train_dmpo --gpus N` re-executes itself under torch.distributed.run with one rank
        per GPU (per-rank shard + replay, one flat gradient all-reduce per learner step over RCCL, hidden behind the next step's
        target-network forwards), while this benchmark's ranks wait on the host.  A failure or a hang of that job costs this leg, not
        the benchmark line.  precision = the physics arithmetic: 64 is the reported figure (the reference's MuJoCo is FP64), 32 is
        reported beside it."""
        import subprocess
        cmd = [sys.executable, '-m', 'flybody_amd.train_dmpo', '--envs', str(args.dmpo_envs), '--iters', str(args.dmpo_iters), '--warmup', str(args.dmpo_warmup),
               '--min-replay', str(args.dmpo_min_replay), '--replay-capacity', str(args.dmpo_replay_capacity), '--precision', str(precision)] + \
              (['--gpus', str(n_ranks)] if n_ranks > 1 else [])
        t0 = time.perf_counter()
        env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'LOCAL_WORLD_SIZE', 'GROUP_RANK',
                                                                 'ROLE_RANK', 'ROLE_WORLD_SIZE', 'GROUP_WORLD_SIZE', 'TORCHELASTIC_RUN_ID', 'TORCHELASTIC_RESTART_COUNT',
                                                                 'TORCHELASTIC_MAX_RESTARTS', 'TORCHELASTIC_USE_AGENT_STORE', 'TORCHELASTIC_ERROR_FILE')}
        import signal
        pr = subprocess.Popen(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, start_new_session=True)
        try:
            so, se = pr.communicate(timeout=args.dmpo_timeout)
        except subprocess.TimeoutExpired:
            try:
                os.killpg(pr.pid, signal.SIGKILL)          # the job's own process group: launcher and ranks (nothing else lives in it)
            except ProcessLookupError:
                pass
            try:
                pr.communicate(timeout=30)
            except subprocess.TimeoutExpired:
                pass
            return {'error': 'timeout'}
        line = [l for l in so.splitlines() if l.startswith('{')]
        if not line:
            return {'error': se[-600:]}
        o = json.loads(line[-1])
        keep = ('n_gpus', 'env_steps_per_sec', 'learner_steps_per_sec', 'envs_per_gpu', 'timed_control_steps', 'wall_s_timed', 'learner_steps_per_env_step',
                'batch_size', 'num_samples', 'replay_capacity', 'min_replay_size', 'samples_per_insert', 'roofline', 'gradient_allreduce', 'dtype', 'physics_build', 'reward')
        out = {k: o[k] for k in keep if k in o}
        out.update({'wall_s_incl_startup': time.perf_counter() - t0,
                    'note': '%d rank(s) x %d environments, physics and learner on the same device(s); %d timed control steps after %d warm-up steps'
                            % (n_ranks, args.dmpo_envs, args.dmpo_iters, args.dmpo_warmup)})
        return out

    stream_pool = []

    def run_split_leg(precision, parts=2, dense=False):
        """The same environments as `parts` independent half-batches, each with its own fb_batch handle and HIP stream: every
        environment still advances K control steps, but the halves are not in lock-step with each other, so the tail of one
        launch (its last long environments) overlaps the head of the other half's next launch.  Secondary number, never `value`."""
        sizes = [n_env // parts + (1 if p < n_env % parts else 0) for p in range(parts)]
        mdl = engine.Model.from_asset('walk_imitation', dense=True) if dense else model
        # (the same stream objects for every pipelined leg: HIP multiplexes streams onto a few hardware queues, and two sub-batches
        # that land on one queue serialise -- measured 22.9 ms instead of 14.2 ms for the three-stream leg)
        while len(stream_pool) < parts:
            stream_pool.append(torch.cuda.Stream())
        streams = stream_pool[:parts]
        batches, actions, bases = [], [], []
        for p in range(parts):
            sub = sizes[p]
            b = engine.Batch(mdl, sub, device=local_rank, precision=precision)
            b.set_reference(qp, qv, terminal_com_dist=float('inf')); b.reset(stream=stream)
            batches.append(b); actions.append(torch.empty(sub, nu, device='cuda', dtype=torch.float32))
            bases.append(id_base + sum(sizes[:p]))
        torch.cuda.synchronize()

        def one_step(t):
            for p in range(parts):
                batches[p].random_actions(actions[p].data_ptr(), t, seed=args.seed, env_id_base=bases[p], stream=streams[p].cuda_stream)
                batches[p].step_ptr(actions[p].data_ptr(), streams[p].cuda_stream)

        for k in range(args.warmup):
            one_step(k)
        barrier()
        t0 = time.perf_counter()
        for k in range(args.steps):
            one_step(args.warmup + k)
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device='cuda' if backend == 'nccl' else 'cpu', dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        finite = all(bool(np.isfinite(b.get('QPOS')).all()) for b in batches)
        del batches
        return dt, finite, sum(sizes)

    extras = {}
    dt, kernel_ms, nlaunch, finite = run_leg(args.precision, extras)
    split = None if args.no_split_leg else run_split_leg(args.precision)
    # ... and on the 12-environments-per-CU build of the same kernel (engine.HIP_LIB_DENSE) as three sub-batches: slower than the
    # default build in lock-step, faster pipelined (more resident environments)
    split_dense = None
    if not args.no_split_leg and args.precision == 64 and os.path.exists(engine.HIP_LIB_DENSE):
        split_dense = run_split_leg(64, parts=3, dense=True)
    f32 = None
    if args.precision == 64 and not args.no_f32_leg:
        f32 = run_leg(32)
    flight = dmpo = None
    if not args.no_secondary_configs:
        if not args.no_flight_leg:
            flight = {f'f{p}': run_flight_leg(p, max(10, args.steps), max(5, args.warmup)) for p in ((64, 32) if args.precision == 64 else (32,))}
        if rank == 0:
            dmpo = run_dmpo_leg(world, 64)              # configs[2] on one rank, configs[4] (its own N-rank job) on several; FP64 physics
            if world == 1 and not args.no_dmpo_f32 and 'error' not in dmpo:
                f = run_dmpo_leg(1, 32)                 # the same loop on the FP32 build of the physics kernel, reported beside it
                dmpo['f32_physics'] = {k: f[k] for k in ('env_steps_per_sec', 'learner_steps_per_sec', 'samples_per_insert', 'roofline', 'dtype', 'error') if k in f}
        if world > 1:                                   # the other ranks wait on the HOST (rendezvous store), not in a device-side collective that would
            from datetime import timedelta              # keep a polling kernel on the GPUs the DMPO job is measured on
            store = dist.distributed_c10d._get_default_store()
            if rank == 0:
                store.set('fb_dmpo_leg_done', '1')
            else:
                # (the job is killed after --dmpo-timeout, the kill itself is bounded by 30 s: wait longer than both, and a rank whose wait
                # still expires reports the leg as lost instead of taking the benchmark line down)
                try:
                    store.wait(['fb_dmpo_leg_done'], timedelta(seconds=args.dmpo_timeout + 180))
                except Exception:
                    pass
        barrier()
    parity = None
    if rank == 0 and args.precision == 64 and not args.no_parity_sample:
        parity = parity_sample(extras)
    traffic = None
    tr_path = os.path.join(ROOT, 'profiles', f'pmc_traffic_f{args.precision}.json')
    if os.path.exists(tr_path):
        # recorded by tools/collect_profiles.sh with rocprofv3 --pmc (separate FETCH_SIZE / WRITE_SIZE passes)
        traffic = json.load(open(tr_path))
    if rank == 0:
        total_env_steps = n_env * world * args.steps
        value = total_env_steps / dt
        per_launch_s = (kernel_ms / 1e3) / max(nlaunch, 1)
        lm = np.sort(np.asarray(extras.get('launch_ms', []), np.float64))
        # every timed launch on its own (HIP events around each): one long control step -- the environment with the largest constraint
        # system ends the launch -- shows here instead of silently moving a 20-step mean
        launch_stats = None if lm.size == 0 else {'n': int(lm.size), 'min': float(lm[0]), 'median': float(np.median(lm)), 'mean': float(lm.mean()),
                                                  'p90': float(lm[min(lm.size - 1, int(np.ceil(0.9*lm.size)) - 1)]), 'max': float(lm[-1])}
        algo_bytes = ALGO_BYTES_PER_ENV_STEP[args.precision]
        achieved_gbs = algo_bytes * n_env / per_launch_s / 1e9
        valu_tflops = ALGO_FLOP_PER_ENV_STEP * n_env / per_launch_s / 1e12
        out = {
            'metric': 'env steps/sec (whole node), walk_imitation 4096-batch random-action rollout',
            'value': value, 'unit': 'env steps/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'preroll': args.preroll,
            'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32' if args.precision == 32 else 'f64', 'data': 'synthetic',
            'config': {'workload': 'configs[1]: 4096 batched walk_imitation envs per GPU, random-action rollout, '
                                   'physics kernels + obs/reward/termination epilogue, no learner',
                       'envs_per_gpu': n_env, 'global_envs': n_env * world, 'substeps_per_step': model.dim('nsubstep'),
                       'parallelism': f'env-shard x{world}, no data-path collective', 'state_finite': finite,
                       'solver': 'Newton (the reference XML sets no solver = MuJoCo default), constraint-space restatement; noslip 3',
                       'solver_iterations_mean': extras.get('solver_iterations_mean'),
                       'auto_resets': extras.get('auto_resets'), 'scheduler': extras.get('scheduler'),
                       'build': ('libflybody_hip_dense.so: FP64, 12 environments per CU (3072 resident)' if dense_headline else 'libflybody_hip.so: default build (FP64: 8 environments per CU, 2048 resident)'),
                       'build_version': engine.version(engine.HIP_LIB_DENSE if dense_headline else None), 'sources_in_tree': engine.source_hash()},
            # the binding roofline of this path is the vector ALU (neither HBM nor MFMA bounds it), so the primary
            # achieved/peak/frac are algorithmic FLOP/s against the vector peak of the arithmetic type; the HBM view the
            # contract also asks for (algorithmic bytes / launch time against 8 TB/s, and the PMC traffic) sits in `hbm`
            'roofline': {'bound': 'valu', 'achieved': valu_tflops, 'peak': VALU_PEAK_TFLOPS[args.precision], 'unit': 'TFLOP/s',
                         'frac': valu_tflops / VALU_PEAK_TFLOPS[args.precision],
                         'traffic': (traffic or {}).get('bytes_per_launch'),
                         'traffic_source': (traffic or {}).get('source'),
                         'kernel': 'k_fly (one control step of all envs)', 'kernel_ms_avg': per_launch_s * 1e3,
                         'kernel_ms': launch_stats,
                         # (staggered pre-roll: every timed launch carries ~n_env/236 resetting environments, none is a reset-only pass)
                         'algorithmic_flop_per_env_step': ALGO_FLOP_PER_ENV_STEP,
                         'algorithmic_bytes_per_env_step': algo_bytes,
                         'hbm': {'achieved': achieved_gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved_gbs / HBM_PEAK_GBS},
                         'note': 'vector-ALU / dependent-latency bound (one environment per wavefront, ONE ROW PER ENVIRONMENT in HBM rather than SoA across '
                                 'environments: every derived array of a substep round-trips through the row, which is where `traffic` comes from -- DESIGN.md 3); '
                                 'flops are the instrumented 3.14 MFLOP/env-step of the FP64 CPU oracle with the Newton solver (tools/flopcount; 3.44 with PGS)',
                         'valu_achieved_tflops': valu_tflops, 'valu_peak_tflops': VALU_PEAK_TFLOPS[args.precision],
                         'valu_frac': valu_tflops / VALU_PEAK_TFLOPS[args.precision]},
            'parity_sample': parity,
            'warn': {'envs_with_flag_since_reset': extras.get('warn'), 'sizes': extras.get('sizes'), 'note': 'FB_WARN_EVER population over the batch: contact cap (64), constraint-row cap (192), '
                     'solver at opt.iterations, MPR at its iteration limit -- MuJoCo reports the first two as nconmax / njmax warnings (fruitfly.xml:6)'},
            'rccl': ('exercised, ranks: %d (checked: all_reduce(SUM) of the rank ids = N(N-1)/2 and all_gather = 0..N-1 on every rank; then all_reduce(MAX) of the timings + barriers)%s' % (world, '; dmpo_mode: one flat gradient all-reduce per learner step' if dmpo is not None else '')) if (world > 1 and backend == 'nccl') else
                    ('unexercised (gloo substitute)' if world > 1 else 'unexercised (single rank: no collective on the data path)'),
            'parity': 'FP64 kernel vs in-repo FP64 C oracle (1e-6 over 100 control steps, tests/test_gpu_parity.py); '
                      'parity vs CPU MuJoCo is UNPINNED (no MuJoCo here; tools/dump_mujoco_golden.py + tests/test_mujoco_golden.py)',
        }
        if f32 is not None:
            out['f32_mode'] = {'value': total_env_steps / f32[0], 'unit': 'env steps/sec', 'ms_per_step': f32[0] / args.steps * 1e3,
                               'kernel_ms_avg': f32[1] / max(f32[2], 1), 'state_finite': f32[3],
                               'note': 'same workload on the FP32 build of the kernel (FP64 residual accumulation in the solver); '
                                       'statistical parity only, see DESIGN.md 6 -- not the headline'}
        if split is not None:
            out['two_stream_mode'] = {'value': split[2] * world * args.steps / split[0], 'unit': 'env steps/sec', 'ms_per_step': split[0] / args.steps * 1e3,
                                      'state_finite': split[1],
                                      'note': 'same environments stepped as 2 independent half-batches (2 fb_batch handles, 2 HIP streams): each half '
                                              'is in lock-step, the halves are not, so one launch\'s tail overlaps the other\'s head; an actor-side '
                                              'scheduling option (tools/split_bench.py), not the headline'}
        if split_dense is not None:
            out['pipelined_dense_mode'] = {'value': split_dense[2] * world * args.steps / split_dense[0], 'unit': 'env steps/sec',
                                           'ms_per_step': split_dense[0] / args.steps * 1e3, 'state_finite': split_dense[1],
                                           'note': '3 independent sub-batches on 3 HIP streams, FB_F64_DENSE build (12 instead of 8 FP64 environments per CU); '
                                                   'secondary like two_stream_mode: the headline is the lock-step batch (config.build says on which build)'}
        if flight is not None:
            out['flight_mode'] = {'config': 'configs[3]: flight_imitation, 8192 envs per GPU, U(-1,1)^12 actions, WBPG + ellipsoid wing fluid forces', **flight}
        if dmpo is not None:
            cfg_name = ('configs[2]: walk_imitation DMPO training, %d envs, on-GPU rollout + learner + replay; reference parameters: replay 4 M, '
                        'min_replay 10 k, SPI 15, batch 256, N 20' % args.dmpo_envs) if world == 1 else \
                       ('configs[4]: walk_imitation DMPO, %d envs sharded across %d GPUs (%d per GPU), gradient all-reduce over %s, SPI 15'
                        % (args.dmpo_envs*world, world, args.dmpo_envs, 'RCCL / xGMI' if backend == 'nccl' else 'gloo (one-GPU functional run)'))
            out['dmpo_mode'] = {'config': cfg_name, **dmpo}
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline()
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
