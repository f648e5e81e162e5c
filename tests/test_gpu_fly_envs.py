"""The dm_env surface of fly_envs.walk_imitation(), mirroring the reference's own env tests
(tests/test_walking_env.py:37-72) on the HIP engine."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

expect_obs_names = ['walker/' + s for s in ['accelerometer', 'actuator_activation', 'appendages_pos', 'force', 'gyro',
                                            'joints_pos', 'joints_vel', 'touch', 'velocimeter', 'world_zaxis',
                                            'ref_displacement', 'ref_root_quat']]
n_steps = 200
qpos = np.zeros((n_steps, 7)); qpos[:, 0] = np.arange(0, n_steps*0.002, 0.002); qpos[:, [2, 3]] = [0.14355, 1.]
qvel = np.zeros((n_steps, 6)); qvel[:, 0] = 1.


def _notebook_specs():
    import json, os
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'notebook_specs.json')))


def test_observation_and_action_specs_match_the_reference_notebooks():
    """observation_spec() keys / order / shapes and action names of the envs the reference's notebooks print
    (docs/getting-started.ipynb cells 44, 46: walk_on_ball; docs/sensory-input-tracking.ipynb cells 8, 9: flight_imitation)."""
    from flybody_amd.fly_envs import walk_on_ball, flight_imitation
    specs = _notebook_specs()
    for env, okey, akey in ((walk_on_ball(precision=64), 'walk_on_ball_observation_spec', 'walk_on_ball_action_spec'),
                            (flight_imitation(precision=64), 'flight_observation_spec', 'flight_action_spec_canonical')):
        obs = env.observation_spec()
        assert list(obs) == specs[okey]['keys']
        assert [list(v.shape) for v in obs.values()] == specs[okey]['shapes']
        assert env.action_spec().name.split('\t') == specs[akey]['names']
        ts = env.reset()
        assert [list(np.shape(ts.observation[k])) for k in specs[okey]['keys']] == specs[okey]['shapes']


def test_can_create_env_inference_mode():
    from flybody_amd.fly_envs import walk_imitation
    env = walk_imitation(terminal_com_dist=float('inf'), precision=64)
    assert list(env.observation_spec()) == expect_obs_names
    spec = env.action_spec()
    assert spec.shape == (59,) and (spec.minimum < spec.maximum).all() and len(spec.name.split('\t')) == 59
    # literally the vector the reference's notebook prints (docs/getting-started.ipynb cell 46 -> tests/golden/notebook_specs.json)
    ref = _notebook_specs()['walk_on_ball_action_spec']
    assert spec.name.split('\t') == ref['names']
    assert np.allclose(spec.minimum, ref['minimum'], rtol=0, atol=1e-12) and np.allclose(spec.maximum, ref['maximum'], rtol=0, atol=1e-12)
    env.task._traj_generator.set_next_trajectory(qpos, qvel)
    ts = env.reset()
    assert ts.first() and ts.reward is None
    for name in expect_obs_names:
        assert isinstance(ts.observation[name], np.ndarray)
    assert ts.observation['walker/ref_displacement'].shape == (65, 3) and ts.observation['walker/ref_root_quat'].shape == (65, 4)
    assert np.isclose(env.control_timestep(), 2e-3) and np.isclose(env.physics.timestep(), 2e-4)


def test_can_step_env_inference_mode():
    from flybody_amd.fly_envs import walk_imitation
    env = walk_imitation(terminal_com_dist=float('inf'), precision=64)
    env.task._traj_generator.set_next_trajectory(qpos, qvel)
    env.reset()
    for _ in range(100):
        ts = env.step(np.random.uniform(-0.5, 0.5, 59))
        assert ts.reward == 1.        # at test time the reward is 1
    assert ts.mid() and ts.discount == 1.0
    # episode ends after min(5001, 200 - 65) = 135 steps, then the next step() is a reset
    for _ in range(35):
        ts = env.step(np.zeros(59))
    assert ts.last() and ts.discount == 1.0
    ts = env.step(np.zeros(59))
    assert ts.first()


def test_batched_tensor_interface():
    import torch
    from flybody_amd.fly_envs import walk_imitation
    env = walk_imitation(n_env=256, precision=32, terminal_com_dist=float('inf'))
    v = env.reset_all()
    assert v['obs'].shape == (256, 741) and v['obs'].is_cuda
    a = torch.zeros(256, 59, device='cuda')
    v = env.step_tensor(a); torch.cuda.synchronize()
    assert torch.isfinite(v['obs']).all() and (v['reward'] == 1).all() and (v['step_type'] == 1).all()
    assert torch.equal(v['obs'][0], v['obs'][255])


def test_dmpo_training_loop_smoke():
    """BASELINE configs[2] plumbing: on-GPU rollout -> n-step replay -> DMPO learner step."""
    import torch
    from flybody_amd.dmpo import DMPOConfig
    from flybody_amd.train_dmpo import Trainer
    tr = Trainer(n_env=256, precision=32, replay_capacity=20_000, config=DMPOConfig(min_replay_size=1000, batch_size=64, num_samples=8),
                 terminal_com_dist=float('inf'))
    stats = None
    for _ in range(12):
        stats = tr.iterate() or stats
    torch.cuda.synchronize()
    assert tr.replay.size > 1000 and tr.learner_steps > 0 and stats is not None
    assert all(torch.isfinite(v).all() for v in stats.values())
    assert torch.isfinite(tr.obs).all()


def test_flight_imitation_env():
    """flight_imitation on the GPU: spec sizes (docs/sensory-input-tracking.ipynb cells 8-9: 104 obs floats, 12 actions),
    reward in (0, 1], episode of 194 steps with a 'good' termination."""
    import torch
    from flybody_amd.fly_envs import flight_imitation
    env = flight_imitation(precision=64)
    assert env.action_spec().shape == (12,) and env.action_spec().name.split('\t')[-1] == 'user_0'
    assert sum(int(np.prod(v.shape)) for v in env.observation_spec().values()) == 104
    assert np.isclose(env.control_timestep(), 2e-4) and np.isclose(env.physics.timestep(), 5e-5)
    ts = env.reset(); assert ts.first()
    for k in range(194):
        ts = env.step(np.zeros(12))
        assert 0.0 <= ts.reward <= 1.0
        if ts.last():
            break
    assert ts.last() and k == 193 and ts.discount == 1.0
    venv = flight_imitation(n_env=512, precision=32)
    v = venv.reset_all()
    a = torch.rand(512, 12, device='cuda') * 2 - 1
    for _ in range(5):
        v = venv.step_tensor(a)
    torch.cuda.synchronize()
    assert torch.isfinite(v['obs']).all() and (v['reward'] > 0).all()


@pytest.mark.gpu
def test_trainer_checkpoint_resume_and_evaluator(tmp_path):
    """Checkpoint / snapshot / metrics of the on-GPU trainer and the greedy evaluator."""
    import glob, json, os
    import torch
    from flybody_amd.train_dmpo import Trainer
    from flybody_amd.dmpo import DMPOConfig, load_policy_snapshot
    cfg = DMPOConfig(min_replay_size=256, batch_size=64, num_samples=4)
    os.environ['FB_LEARNER_GRAPHS'] = '0'
    tr = Trainer(n_env=64, precision=32, replay_capacity=20_000, config=cfg, terminal_com_dist=float('inf'), directory=str(tmp_path),
                 time_delta_minutes=1e9)
    for _ in range(24):
        tr.iterate()
    assert tr.learner.num_steps > 0
    assert tr.checkpointer.save(force=True) and tr.snapshotter.save(force=True, actor_steps=int(tr.counter.counts['actor_steps']))
    m = tr.log()
    assert m['actor_steps'] == 24*64 and m['learner_steps'] == tr.learner.num_steps and 'steps_per_second_actor' in m
    ev = tr.evaluate(n_env=8, episodes_per_env=1)
    assert ev['episodes'] >= 8 and ev['episode_length'] > 1 and ev['episode_return'] > 0
    steps = tr.learner.num_steps
    w0 = [p.detach().clone() for p in tr.learner.online.parameters()]
    del tr
    tr2 = Trainer(n_env=64, precision=32, replay_capacity=20_000, config=cfg, terminal_com_dist=float('inf'), directory=str(tmp_path))
    assert tr2.restored_from is not None and tr2.learner.num_steps == steps and tr2.counter.counts['actor_steps'] == 24*64
    for a, b in zip(w0, tr2.learner.online.parameters()):
        assert torch.equal(a, b)
    pol, meta = load_policy_snapshot(glob.glob(str(tmp_path / 'snapshots' / 'policy-*.pt'))[0], device='cuda')
    assert meta['obs_dim'] == tr2.env.nobs and meta['action_dim'] == 59
    assert json.loads(open(tmp_path / 'metrics_learner.jsonl').read().splitlines()[-1])['actor_steps'] == 24*64


@pytest.mark.gpu
def test_ellipsoid_fluid_force_analysis_api():
    """flybody_amd.fluid.ellipsoid_fluid_forces (counterpart of flybody/ellipsoid_fluid_model.py:16-78) on a flapping fly:
    components in global coordinates and qfrc_fluid against the CPU oracle evaluated at the same state."""
    import ctypes as C, os
    from conftest import ROOT
    from flybody_amd.fly_envs import flight_imitation
    from flybody_amd.fluid import ellipsoid_fluid_forces, COMPONENTS
    from flybody_amd.model_blob import load_npz, pack_model
    from oracle import fbo
    env = flight_imitation(precision=64)
    env.reset()
    rng = np.random.default_rng(0)
    for _ in range(12):
        env.step(rng.uniform(-0.5, 0.5, 12))
    forces, qfrc_fluid = ellipsoid_fluid_forces(env)
    assert set(forces) == {'wing_left', 'wing_right'}
    arr = load_npz(os.path.join(ROOT, 'flybody_amd', 'assets', 'flight_imitation.npz'))
    od = fbo.OracleData(fbo.OracleModel(pack_model(arr)))
    od.field('qpos')[:] = env.batch.get('QPOS')[0]; od.field('qvel')[:] = env.batch.get('QVEL')[0]
    od.call('fwd_position'); od.call('fwd_velocity')
    assert np.allclose(qfrc_fluid, od.field('qfrc_fluid'), rtol=1e-7, atol=1e-12)
    L = fbo.lib()
    L.fbo_ellipsoid_local.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
    gxp = od.field('geom_xpos').reshape(-1, 3); gxm = od.field('geom_xmat').reshape(-1, 3, 3)
    cvel = od.field('cvel').reshape(-1, 6); com = od.field('subtree_com')[3:6]
    for body, geoms in forces.items():
        for g, comp in geoms.items():
            b = int(arr['geom_bodyid'][g])
            ang, lin = cvel[b, :3], cvel[b, 3:] - np.cross(gxp[g] - com, cvel[b, :3])
            lvel = np.ascontiguousarray(np.concatenate([gxm[g].T @ ang, gxm[g].T @ lin]))
            lfrc = np.zeros(6); cc = np.zeros(24)
            size = np.ascontiguousarray(arr['geom_size'][g]); gf = np.ascontiguousarray(arr['geom_fluid'][g])
            L.fbo_ellipsoid_local(lvel.ctypes.data, size.ctypes.data, gf.ctypes.data, float(arr['opt_density']), float(arr['opt_viscosity']),
                                  lfrc.ctypes.data, cc.ctypes.data)
            for k, name in enumerate(COMPONENTS):
                assert np.allclose(comp[name], gxm[g] @ (cc[3*k:3*k + 3]*gf[0]), rtol=1e-6, atol=1e-14), (body, g, name)
            total_f = sum(comp[n] for n in COMPONENTS if n.startswith('f')); total_g = sum(comp[n] for n in COMPONENTS if n.startswith('g'))
            assert np.allclose(total_f, gxm[g] @ lfrc[3:], rtol=1e-6, atol=1e-14) and np.allclose(total_g, gxm[g] @ lfrc[:3], rtol=1e-6, atol=1e-14)
            assert np.linalg.norm(total_f) > 0


@pytest.mark.gpu
def test_factory_variants_on_gpu():
    """FruitFly._build's switches through the fly_envs factories (force_actuators, enabled wings / legs, unfiltered joints):
    spec sizes follow the configuration and the environments step (the compiled variants are committed under assets/variants)."""
    import torch
    from flybody_amd.fly_envs import flight_imitation, walk_imitation, walk_on_ball
    e = walk_imitation(force_actuators=True, terminal_com_dist=float('inf'), n_env=64, precision=32)
    spec = e.action_spec()
    names = spec.name.split('\t')
    lo, hi = np.asarray(spec.minimum), np.asarray(spec.maximum)
    adh = np.array(['adhere' in n for n in names])
    assert spec.shape == (59,) and np.all(lo[~adh] == -1) and np.all(hi[~adh] == 1) and np.all(lo[adh] == 0) and np.all(hi[adh] == 1)
    v = e.reset_all(); a = torch.rand(64, 59, device='cuda')*2 - 1
    for _ in range(3):
        v = e.step_tensor(a)
    torch.cuda.synchronize(); assert torch.isfinite(v['obs']).all()
    e = walk_imitation(disable_wings=False, terminal_com_dist=float('inf'), n_env=8, precision=64)
    assert e.action_spec().shape == (65,) and sum(int(np.prod(s.shape)) for s in e.observation_spec().values()) == 741 + 12 + 6     # 6 more joints (pos, vel) and activations
    ts = e.reset(); ts = e.step(np.zeros(65)); assert ts.reward == 1.0
    e = walk_imitation(joint_filter=0.0, terminal_com_dist=float('inf'), n_env=8, precision=64)
    assert e.observation_spec()['walker/actuator_activation'].shape == (6,)          # only the adhesion filters keep a state
    ts = e.reset(); ts = e.step(np.zeros(59)); assert np.isfinite(ts.observation['walker/joints_pos']).all()
    e = flight_imitation(disable_legs=False, n_env=8, precision=64)
    assert e.action_spec().shape == (66,) and e.action_spec().name.split('\t')[-1] == 'user_0'
    ts = e.reset(); ts = e.step(np.zeros(66)); assert 0.0 <= ts.reward <= 1.0
    e = walk_on_ball(force_actuators=True, n_env=8, precision=32)
    ts = e.reset(); ts = e.step(np.zeros(59)); assert np.isfinite(ts.observation['walker/ball_qvel']).all()


@pytest.mark.gpu
def test_flight_dataset_on_gpu(tmp_path):
    """flight_imitation(ref_path=...) with the dataset resident on the GPU: per-environment trajectory / start-step selection and
    a tracked episode against the oracle (the emulation version is tests/test_flight_dataset.py)."""
    import torch
    from _synthetic_flight_dataset import make_flight_dataset
    from flybody_amd.fly_envs import flight_imitation
    from flybody_amd.model_blob import pack_model
    from flybody_amd.wbpg import build_tables
    from oracle import fbo
    ds = make_flight_dataset(); path = str(tmp_path / 'flight.npz'); ds.save(path)
    n = 16
    env = flight_imitation(ref_path=path, n_env=n, precision=64, seed=11, env_id_base=40)
    arrays = env.model.arrays
    root = ds.root_qpos(arrays['com_offset']); tabs = build_tables()
    om = fbo.OracleModel(pack_model(arrays)); ods = []
    for e in range(n):
        od = fbo.OracleData(om); od.set_wbpg(tabs, seed=11)
        od.set_flight_dataset(ds.offsets, root, ds.com_qvel, future_steps=5, terminal_com_dist=2.0, time_limit=0.6, randomize_start_step=True,
                              seed=11, env_id=40 + e)
        od.env_reset(); ods.append(od)
    v = env.reset_all(); torch.cuda.synchronize()
    obs = v['obs'].cpu().numpy()
    offs = {int(od.scalar('ds_off')) for od in ods}
    assert len(offs) > 4                                                     # the environments picked different slices
    for e, od in enumerate(ods):
        assert np.allclose(obs[e], od.field('obs'), rtol=1e-5, atol=1e-5), e
    rng = np.random.default_rng(1)
    for k in range(40):
        a = rng.uniform(-0.3, 0.3, (n, 12)).astype(np.float32)
        v = env.step_tensor(torch.from_numpy(a).cuda()); torch.cuda.synchronize()
        for e, od in enumerate(ods):
            od.env_step(a[e].astype(np.float64))
        assert v['step_type'].cpu().numpy().tolist() == [int(od.scalar('step_type')) for od in ods], k
    Q = env.batch.get('QPOS')
    for e, od in enumerate(ods):
        assert np.abs(Q[e] - od.field('qpos')).max() < 1e-7*max(1.0, np.abs(od.field('qpos')).max()), e
    assert np.allclose(v['reward'].cpu().numpy(), [od.scalar('reward') for od in ods], atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize('graphs', ['1', '0'])
def test_dmpo_two_ranks_stay_identical(graphs, tmp_path):
    """BASELINE configs[4] plumbing on ONE GPU: two ranks (gloo, both on cuda:0), per-rank environment shard + replay, one flat
    gradient all-reduce per learner step between the forward/backward graph and the optimizer graph; replicas stay identical."""
    import os, socket, subprocess, sys
    from conftest import ROOT
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]
    env = dict(os.environ, FB_BENCH_DEVICE='0', FB_BENCH_BACKEND='gloo', MASTER_ADDR='127.0.0.1', FB_LEARNER_GRAPHS=graphs)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'tests', '_dmpo_two_ranks.py')]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and 'TWO_RANKS_OK' in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
    if graphs == '1':
        # the PIPELINED step ran (target-network forwards of step t + 1 on a side stream against B / all-reduce / OPT of step t,
        # learner.py _step_pipelined) -- and the serial order (FB_LEARNER_PIPELINE=0: one forward/backward graph, all-reduce, optimizer
        # graph) ends with the same parameters.  Not bit for bit across RUNS: the backward kernels accumulate column sums with float
        # atomics, whose order varies from run to run (measured between two runs of one mode: ~1e-8); a pipeline bug -- a stale batch,
        # a half-written buffer -- moves parameters by the learning-rate scale (1e-4 per update, 20 updates).
        import numpy as np
        assert 'pipelined=True' in r.stdout, r.stdout[-500:]
        outs = []
        for mode, extra in (('pipe', {}), ('pipe2', {}), ('serial', {'FB_LEARNER_PIPELINE': '0'})):
            with socket.socket() as s:
                s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]
            cmd[cmd.index('--master-port') + 1] = str(port)
            path = os.path.join(str(tmp_path), mode + '.npy')
            r2 = subprocess.run(cmd, cwd=ROOT, env=dict(env, FB_TEST_PARAMS_OUT=path, **extra), capture_output=True, text=True, timeout=900)
            assert r2.returncode == 0 and ('pipelined=%s' % (mode != 'serial')) in r2.stdout, (r2.stdout[-1500:], r2.stderr[-3000:])
            outs.append(np.load(path))
        # (relative: the dual variables are O(100 ... 1000), one float32 ulp there is 6e-5)
        rel = lambda x, y: float((np.abs(x - y)/np.maximum(np.abs(y), 1.0)).max())
        noise, diff = rel(outs[0], outs[1]), rel(outs[0], outs[2])
        print('two runs of the pipelined step differ by %.2e (relative), pipelined vs serial by %.2e' % (noise, diff))
        assert noise < 2e-6 and diff < 2e-6, (noise, diff)


@pytest.mark.gpu
def test_two_half_batches_on_two_streams_equal_one_batch():
    """bench.py's two_stream_mode / tools/split_bench.py: the same environments stepped as two fb_batch handles on two HIP streams
    (their launches overlap) give bit-identical observations to one batch in lock-step -- environments are independent."""
    import torch
    from flybody_amd.fly_envs import walk_imitation
    n = 512
    one = walk_imitation(n_env=n, precision=64, terminal_com_dist=float('inf'))
    halves = [walk_imitation(n_env=n//2, precision=64, terminal_com_dist=float('inf'), env_id_base=k*n//2) for k in range(2)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    one.reset_all()
    for h in halves:
        h.reset_all()
    torch.cuda.synchronize()
    g = torch.Generator(device='cuda'); g.manual_seed(3)
    for step in range(12):
        a = torch.empty(n, 59, device='cuda').normal_(generator=g).clamp_(-1, 1)
        v1 = one.step_tensor(a)
        torch.cuda.synchronize()
        outs = []
        for k, h in enumerate(halves):
            with torch.cuda.stream(streams[k]):
                outs.append(h.step_tensor(a[k*n//2:(k + 1)*n//2].contiguous()))
        torch.cuda.synchronize()
        assert torch.equal(torch.cat([o['obs'] for o in outs]), v1['obs']), step
        assert torch.equal(torch.cat([o['reward'].view(-1) for o in outs]), v1['reward'].view(-1))


@pytest.mark.gpu
def test_dmpo_two_ranks_identical_across_a_target_sync():
    """The overlapped data-parallel step (learner.py _step_pipelined: next step's target-network forwards on a side stream, the flat
    gradient all-reduce between the backward graph and the optimizer graph) over >= 120 learner steps, i.e. ACROSS the target-network
    copies at steps 101 (policy) and 107 (critic) -- the one place where phase A of the next step must NOT run ahead.  Two ranks on
    one GPU over gloo; online parameters, duals and target networks bit-equal on both ranks afterwards."""
    import os, socket, subprocess, sys
    from conftest import ROOT
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]
    env = dict(os.environ, FB_BENCH_DEVICE='0', FB_BENCH_BACKEND='gloo', MASTER_ADDR='127.0.0.1', FB_LEARNER_GRAPHS='1', FB_TEST_LSTEPS='8', FB_TEST_ITERS='26')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'tests', '_dmpo_two_ranks.py')]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and 'TWO_RANKS_OK' in r.stdout and 'pipelined=True' in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
    import re
    steps = int(re.search(r'steps=(\d+)', r.stdout).group(1))
    assert steps >= 120, steps
