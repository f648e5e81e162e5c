"""Compiled-model constants against the values the reference's own tests pin
(tests/golden/reference_pins.json cites reference file:line for each)."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PINS = json.load(open(os.path.join(HERE, 'golden', 'reference_pins.json')))


def _subtree(arrays, name):
    return float(arrays['body_subtreemass'][list(arrays['names_body']).index(name)])


def test_hot_path_dimensions(walk_arrays):
    a = walk_arrays; p = PINS['hot_path_dims']['walk']
    assert len(a['qpos0']) == p['nq'] and len(a['dof_bodyid']) == p['nv']
    assert len(a['actuator_trntype']) == p['nu'] == PINS['walk_env']['num_actions']
    assert int((a['actuator_actadr'] >= 0).sum()) == p['na']
    assert len(a['body_parent']) == PINS['dims_full_model']['nbody']       # world + 67 bodies
    assert len(a['jnt_type']) == PINS['dims_full_model']['njnt']
    assert len(a['tendon_adr']) == PINS['dims_full_model']['ntendon']
    assert len(a['site_bodyid']) == PINS['dims_full_model']['nsite']
    t1 = a['geom_type'][a['pair_geom1']]
    assert int((t1 == 0).sum()) == p['floor_pairs'] and int((t1 != 0).sum()) == p['self_pairs']
    assert float(a['opt_timestep']) == PINS['walk_env']['physics_timestep']
    assert float(a['opt_control_timestep']) == PINS['walk_env']['control_timestep']


def test_masses_match_reference_pins(walk_arrays):
    m = PINS['masses']; a = walk_arrays
    assert np.isclose(_subtree(a, 'thorax'), m['fly_mass'])
    assert np.isclose(_subtree(a, 'head'), m['head'])
    assert np.isclose(a['body_mass'][list(a['names_body']).index('thorax')], m['thorax'])
    assert np.isclose(_subtree(a, 'abdomen'), m['abdomen'])
    for side in ('left', 'right'):
        assert np.isclose(_subtree(a, f'coxa_T1_{side}'), m['leg_T1'])
        assert np.isclose(_subtree(a, f'coxa_T2_{side}'), m['leg_T2'])
        assert np.isclose(_subtree(a, f'coxa_T3_{side}'), m['leg_T3'])
        assert np.isclose(a['body_mass'][list(a['names_body']).index(f'wing_{side}')], m['wing'])


def test_action_to_ctrl_map_and_ranges(walk_arrays):
    """Action order adhesion, head, abdomen, legs (fruitfly.py:25-32,342-379); position actuators'
    ctrl range equals the joint range (tests/test_flybare.py:76-88)."""
    a = walk_arrays
    names = [str(n) for n in a['names_actuator']]
    order = [names[i] for i in a['action_to_ctrl']]
    assert all('adhere' in n for n in order[:6])
    assert order[6:9] == ['head_abduct', 'head_twist', 'head']
    assert order[9:11] == ['abdomen_abduct', 'abdomen']
    assert all(any(t in n for t in ('T1', 'T2', 'T3')) for n in order[11:])
    assert sorted(a['action_to_ctrl'].tolist()) == list(range(59))
    jn = [str(n) for n in a['names_jnt']]
    for i, n in enumerate(names):
        if a['actuator_trntype'][i] == 0 and a['actuator_biastype'][i] == 1:
            j = a['actuator_trnid'][i]
            assert jn[j] == n
            assert np.array_equal(a['actuator_ctrlrange'][i], a['jnt_range'][j])
    # filter dynamics on every actuator: joint 0.01, adhesion 0.007 (fly_envs.py:106, base.py:37)
    adh = a['actuator_trntype'] == 5
    assert np.allclose(a['actuator_dynprm'][~adh], 0.01) and np.allclose(a['actuator_dynprm'][adh], 0.007)
    assert (a['actuator_dyntype'] == 2).all()


def test_flight_model_dimensions():
    from flybody_amd.model_blob import load_npz
    a = load_npz(os.path.join(os.path.dirname(HERE), 'flybody_amd', 'assets', 'flight_imitation.npz'))
    p = PINS['hot_path_dims']['flight']
    assert len(a['qpos0']) == p['nq'] and len(a['dof_bodyid']) == p['nv'] and len(a['actuator_trntype']) == p['nu']
    assert (a['geom_fluid'][:, 0] > 0).sum() == 2            # the two wing fluid ellipsoids (base.py:319-322)
    # sphere limit of the added-mass integrals: kappa = 2/3 -> virtual mass = V/2
    from flybody_amd.mjcf_compile import ellipsoid_virtual_inertia
    vm, vi = ellipsoid_virtual_inertia([0.1, 0.1, 0.1])
    assert np.allclose(vm, 0.5*4/3*np.pi*1e-3, rtol=1e-6) and np.allclose(vi, 0, atol=1e-12)


def test_observation_layout_matches_reference(walk_arrays):
    from flybody_amd.fly_envs import _DICT_ORDER
    assert [k for k in _DICT_ORDER if k != 'ball_qvel'] == PINS['walk_env']['obs_names']     # ball_qvel: walk_on_ball only
    a = walk_arrays
    nobs = 3 + 59 + 3*len(a['appendage_sites']) + 3*len(a['sensor_force_sites']) + 3 + 2*len(a['observable_joints']) + 65*7 + len(a['sensor_touch_sites']) + 3 + 3
    assert nobs == PINS['hot_path_dims']['walk']['nobs']
    assert len(a['observable_joints']) == 85
