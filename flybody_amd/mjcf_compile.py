"""Offline model compiler: fruitfly.xml (+ task rewrites) -> flat constant tables.

This is a from-scratch MJCF subset compiler written for exactly the model class
of the flybody fruit fly.  It does not depend on MuJoCo or dm_control.  It
restates, for this model only:

  * MJCF default-class resolution and body/geom/joint/site/actuator parsing
    (reference model: flybody/fruitfly/assets/fruitfly.xml:1-918),
  * the config-time rewrites of FruitFly._build (flybody/fruitfly/fruitfly.py:174-381),
    FruitFlyTask/Walking/Flying.__init__ (flybody/tasks/base.py:129-167,296-364,385-411)
    and WalkImitation.__init__ (flybody/tasks/walk_imitation.py:69-73),
  * MuJoCo's compile-time derived quantities that the step needs (body inertial
    frames from mesh/primitive geoms, springdamper, invweight0, collision-pair
    filtering and contact-parameter mixing, fluid-ellipsoid virtual inertia).

The output is a dict of numpy arrays ("compiled model") that is serialised to
``flybody_amd/assets/*.npz``.  Both the CPU oracle (oracle/) and the HIP engine
(flybody_amd/csrc/) consume the same tables through the C-ABI ``fb_model_load``.

Known, declared approximations (see DESIGN.md "Model constants"):
  * 6 mesh files are absent from the reference mount (.MISSING_LARGE_BLOBS:3-8).
    thorax: mass is explicit in the XML (fruitfly.xml:322); its inertia tensor and
    centre of mass are taken from its two collision ellipsoids scaled to that mass.
    head: mass is recovered from the test-pinned head-subtree mass
    (tests/test_flybare.py:29) minus the computable child masses; inertia tensor
    from its collision ellipsoid scaled to that mass.
"""
from __future__ import annotations

import math
import os
import xml.etree.ElementTree as ET
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

# ----------------------------------------------------------------------------
# enums shared with include/flybody_engine.h
JNT_FREE, JNT_BALL, JNT_HINGE = 0, 1, 3
GEOM_PLANE, GEOM_SPHERE, GEOM_CAPSULE, GEOM_ELLIPSOID, GEOM_CYLINDER, GEOM_BOX = 0, 2, 3, 4, 5, 6
TRN_JOINT, TRN_TENDON, TRN_BODY = 0, 3, 5
DYN_NONE, DYN_FILTER, DYN_FILTEREXACT = 0, 2, 3
BIAS_NONE, BIAS_AFFINE = 0, 1
MINVAL = 1e-15

_GEOM_TYPES = {'plane': GEOM_PLANE, 'sphere': GEOM_SPHERE, 'capsule': GEOM_CAPSULE,
               'ellipsoid': GEOM_ELLIPSOID, 'cylinder': GEOM_CYLINDER, 'box': GEOM_BOX,
               'mesh': 7}

HEAD_SUBTREE_MASS_PIN = 0.0001499089219064366   # tests/test_flybare.py:29


# ----------------------------------------------------------------------------
# small quaternion / rotation helpers (w, x, y, z)
def qmul(a, b):
    a = np.asarray(a, float); b = np.asarray(b, float)
    return np.array([
        a[0]*b[0] - a[1]*b[1] - a[2]*b[2] - a[3]*b[3],
        a[0]*b[1] + a[1]*b[0] + a[2]*b[3] - a[3]*b[2],
        a[0]*b[2] - a[1]*b[3] + a[2]*b[0] + a[3]*b[1],
        a[0]*b[3] + a[1]*b[2] - a[2]*b[1] + a[3]*b[0]])


def qconj(q):
    q = np.asarray(q, float)
    return np.array([q[0], -q[1], -q[2], -q[3]])


def qnorm(q):
    q = np.asarray(q, float)
    n = np.linalg.norm(q)
    return q / n if n > 0 else np.array([1., 0, 0, 0])


def q2mat(q):
    w, x, y, z = q
    return np.array([
        [w*w + x*x - y*y - z*z, 2*(x*y - w*z), 2*(x*z + w*y)],
        [2*(x*y + w*z), w*w - x*x + y*y - z*z, 2*(y*z - w*x)],
        [2*(x*z - w*y), 2*(y*z + w*x), w*w - x*x - y*y + z*z]])


def mat2q(m):
    # robust rotation-matrix -> quaternion
    t = np.trace(m)
    if t > 0:
        s = math.sqrt(t + 1.0) * 2
        q = np.array([0.25*s, (m[2, 1]-m[1, 2])/s, (m[0, 2]-m[2, 0])/s, (m[1, 0]-m[0, 1])/s])
    elif m[0, 0] > m[1, 1] and m[0, 0] > m[2, 2]:
        s = math.sqrt(1.0 + m[0, 0] - m[1, 1] - m[2, 2]) * 2
        q = np.array([(m[2, 1]-m[1, 2])/s, 0.25*s, (m[0, 1]+m[1, 0])/s, (m[0, 2]+m[2, 0])/s])
    elif m[1, 1] > m[2, 2]:
        s = math.sqrt(1.0 + m[1, 1] - m[0, 0] - m[2, 2]) * 2
        q = np.array([(m[0, 2]-m[2, 0])/s, (m[0, 1]+m[1, 0])/s, 0.25*s, (m[1, 2]+m[2, 1])/s])
    else:
        s = math.sqrt(1.0 + m[2, 2] - m[0, 0] - m[1, 1]) * 2
        q = np.array([(m[1, 0]-m[0, 1])/s, (m[0, 2]+m[2, 0])/s, (m[1, 2]+m[2, 1])/s, 0.25*s])
    return qnorm(q)


def qrot(q, v):
    return q2mat(q) @ np.asarray(v, float)


def axisangle2q(axis, ang):
    axis = np.asarray(axis, float)
    return np.concatenate(([math.cos(ang/2)], math.sin(ang/2) * axis))


def z2quat(vec):
    """Quaternion rotating the z axis onto ``vec`` (fromto geoms)."""
    vec = np.asarray(vec, float)
    vec = vec / np.linalg.norm(vec)
    z = np.array([0., 0, 1])
    ax = np.cross(z, vec)
    s = np.linalg.norm(ax)
    if s < 1e-10:
        ax = np.array([1., 0, 0])
    else:
        ax = ax / s
    ang = math.atan2(s, vec[2])
    return qnorm(axisangle2q(ax, ang))


def euler2q(e):
    # MuJoCo default eulerseq "xyz", intrinsic rotations
    q = np.array([1., 0, 0, 0])
    for i, ang in enumerate(e):
        ax = np.zeros(3); ax[i] = 1
        q = qmul(q, axisangle2q(ax, ang))
    return q


def _floats(s):
    return np.array([float(x) for x in s.split()], float)


# ----------------------------------------------------------------------------
# mesh mass properties ("legacy" MuJoCo mesh inertia: two passes, |volume|)
def load_obj(path, scale=0.1):
    V, F = [], []
    with open(path) as fh:
        for l in fh:
            if l.startswith('v '):
                V.append([float(x) for x in l.split()[1:4]])
            elif l.startswith('f '):
                idx = [int(t.split('/')[0]) - 1 for t in l.split()[1:]]
                for k in range(1, len(idx) - 1):
                    F.append([idx[0], idx[k], idx[k + 1]])
    return np.array(V) * scale, np.array(F, int)


def mesh_props(V, F):
    """Returns (volume, com[3], inertia[3,3] about com per unit density).

    Pass 1: apex = area-weighted mean of face centroids; |tet volume|-weighted
    centre of mass.  Pass 2: re-centre at that CoM, recompute |tet volumes| and
    the second moments.  (Checked against the test-pinned leg masses,
    tests/test_flybare.py:32-34, to 2e-8 relative.)
    """
    a, b, c = V[F[:, 0]], V[F[:, 1]], V[F[:, 2]]
    n = np.cross(b - a, c - a)
    area = np.linalg.norm(n, axis=1) / 2
    cen = (a + b + c) / 3
    facecen = (cen * area[:, None]).sum(0) / area.sum()
    a0, b0, c0 = a - facecen, b - facecen, c - facecen
    vol = np.abs(np.einsum('ij,ij->i', a0, np.cross(b0, c0))) / 6
    com = facecen + (((a0 + b0 + c0) / 4) * vol[:, None]).sum(0) / vol.sum()
    a1, b1, c1 = a - com, b - com, c - com
    vol = np.abs(np.einsum('ij,ij->i', a1, np.cross(b1, c1))) / 6
    volume = vol.sum()
    # second-moment matrix of a tetrahedron (origin, a, b, c):
    # P = vol/20 * (sum_i v_i v_i^T + (sum_i v_i)(sum_i v_i)^T)
    s = a1 + b1 + c1
    P = (np.einsum('f,fi,fj->ij', vol, a1, a1) + np.einsum('f,fi,fj->ij', vol, b1, b1) +
         np.einsum('f,fi,fj->ij', vol, c1, c1) + np.einsum('f,fi,fj->ij', vol, s, s)) / 20
    I = np.trace(P) * np.eye(3) - P
    return volume, com, I


# ----------------------------------------------------------------------------
@dataclass
class Body:
    name: str
    parent: int
    pos: np.ndarray
    quat: np.ndarray
    childclass: Optional[str]
    joints: List[dict] = field(default_factory=list)
    geoms: List[dict] = field(default_factory=list)
    sites: List[dict] = field(default_factory=list)


class Defaults:
    """MJCF <default> tree: class name -> {tag -> attribute dict}."""

    _ACT_TAGS = ('general', 'adhesion', 'motor', 'position')

    def __init__(self, root_default: Optional[ET.Element]):
        self.classes: Dict[str, Dict[str, Dict[str, str]]] = {}
        self.parent: Dict[str, Optional[str]] = {}
        if root_default is not None:
            self._walk(root_default, 'main', None)
        else:
            self.classes['main'] = {}
            self.parent['main'] = None

    def _walk(self, el, name, parent):
        tags: Dict[str, Dict[str, str]] = {}
        if parent is not None:
            for t, d in self.classes[parent].items():
                tags[t] = dict(d)
        for ch in el:
            if ch.tag == 'default':
                continue
            tag = 'actuator' if ch.tag in self._ACT_TAGS else ch.tag
            attrs = dict(ch.attrib)
            if ch.tag == 'adhesion' and 'gain' in attrs:
                attrs['gainprm'] = attrs.pop('gain')
            tags.setdefault(tag, {}).update(attrs)
        self.classes[name] = tags
        self.parent[name] = parent
        for ch in el:
            if ch.tag == 'default':
                self._walk(ch, ch.attrib['class'], name)

    def get(self, cls: Optional[str], tag: str) -> Dict[str, str]:
        cls = cls or 'main'
        return dict(self.classes[cls].get(tag, {}))

    def set(self, cls: str, tag: str, **attrs):
        """Set attributes on a default class and all descendants that have not overridden them."""
        def rec(c):
            self.classes[c].setdefault(tag, {}).update({k: str(v) for k, v in attrs.items()})
            for k, p in self.parent.items():
                if p == c:
                    rec(k)
        rec(cls)


@dataclass
class TaskConfig:
    """Which rewrites to apply (mirrors FruitFly._build / task kwargs)."""
    name: str = 'walk_imitation'
    use_legs: bool = True
    use_wings: bool = False
    use_mouth: bool = False
    use_antennae: bool = False
    joint_filter: float = 0.01
    adhesion_filter: float = 0.007
    dyntype_filterexact: bool = False
    force_actuators: bool = False      # fruitfly.py:308-327: position actuators -> force actuators (no affine bias, ctrlrange (-1, 1))
    physics_timestep: float = 2e-4
    control_timestep: float = 2e-3
    floor: bool = True                 # dm_control floors.Floor() plane at z=0
    floor_contacts: bool = True        # Flying disables floor contacts (base.py:308-311)
    floor_friction: float = 0.5        # base.py:398-401
    floor_solref: Tuple[float, float] = (0.001, 1.0)
    floor_solimp: Tuple[float, float, float] = (0.95, 0.99, 0.01)
    claw_friction: Optional[float] = 1.0   # walk_imitation.py:69-73
    wing_leg_excludes: bool = True     # base.py:404-411
    body_pitch_angle: float = 47.5
    stroke_plane_angle: float = 0.0
    wing_gainprm: Optional[float] = None       # Flying: 18 (base.py:314-316)
    wing_stiffness: Optional[float] = None     # Flying: 0.01
    wing_damping: Optional[float] = None       # Flying: 0.007769230
    fluidcoef: Optional[Tuple[float, ...]] = None  # Flying: ellipsoid fluid on *fluid* geoms
    num_user_actions: int = 0
    spawn_pos: Tuple[float, float, float] = (0.0, 0.0, 0.1278)   # fruitfly.py:23
    tethered: bool = False             # walk_on_ball.py:28-30: the attachment frame's free joint is removed
    ball: Optional[Tuple[Tuple[float, float, float], float, float]] = None   # arenas/ball.py: (pos, radius, density)
    thorax_child_excludes: bool = False   # walk_on_ball.py:32-40


def walk_imitation_config(joint_filter: float = 0.01) -> TaskConfig:
    return TaskConfig(name='walk_imitation', joint_filter=joint_filter)


def walk_on_ball_config() -> TaskConfig:
    # fly_envs.py:158-191, tasks/walk_on_ball.py:15-48, tasks/arenas/ball.py
    return TaskConfig(name='walk_on_ball', floor=False, tethered=True, ball=((-0.05, 0.0, -0.419), 0.454, 0.0025),
                      thorax_child_excludes=True)


def flight_imitation_config(joint_filter: float = 0.0) -> TaskConfig:
    # fly_envs.py:30-97, tasks/base.py:274-364, constants.py:15-31
    return TaskConfig(name='flight_imitation', use_legs=False, use_wings=True,
                      joint_filter=joint_filter, physics_timestep=5e-5, control_timestep=2e-4,
                      floor_contacts=False, claw_friction=None, wing_leg_excludes=False,
                      wing_gainprm=18.0, wing_stiffness=0.01, wing_damping=0.007769230,
                      fluidcoef=(1.0, 0.5, 1.5, 1.7, 1.0), num_user_actions=1)


_NAME_SUBSTR = {
    'adhesion': [], 'head': ['head'], 'mouth': ['rostrum', 'haustellum', 'labrum'],
    'antennae': ['antenna'], 'wings': ['wing'], 'abdomen': ['abdomen'],
    'legs': ['T1', 'T2', 'T3'], 'user': []}
_ACTION_CLASSES = ['adhesion', 'head', 'mouth', 'antennae', 'wings', 'abdomen', 'legs', 'user']


def _any_in(subs, s):
    return any(x in s for x in subs)


# ----------------------------------------------------------------------------
class FlyCompiler:
    def __init__(self, xml_path: str, cfg: TaskConfig):
        self.cfg = cfg
        self.xml_path = xml_path
        self.asset_dir = os.path.dirname(xml_path)
        self.root = ET.parse(xml_path).getroot()
        self.defaults = Defaults(self.root.find('default'))
        self.meshfile = {m.attrib['name']: m.attrib['file'] for m in self.root.find('asset').findall('mesh')}
        mdef = self.defaults.get('main', 'mesh')
        self.mesh_scale = float(mdef.get('scale', '1 1 1').split()[0])
        self.missing_meshes: List[str] = []
        self.notes: List[str] = []

    # -- attribute resolution ------------------------------------------------
    def _attrs(self, el: ET.Element, tag: str, childclass: Optional[str]) -> Dict[str, str]:
        cls = el.attrib.get('class', childclass)
        d = self.defaults.get(cls, tag)
        for k, v in el.attrib.items():
            if k == 'gain' and tag == 'actuator':
                d['gainprm'] = v
            else:
                d[k] = v
        d['_class'] = cls or 'main'
        return d

    # -- body tree -------------------------------------------------------------
    def parse_bodies(self):
        self.bodies: List[Body] = [Body('world', -1, np.zeros(3), np.array([1., 0, 0, 0]), None)]
        wb = self.root.find('worldbody')

        def rec(el, parent, childclass):
            cc = el.attrib.get('childclass', childclass)
            pos = _floats(el.attrib.get('pos', '0 0 0'))
            quat = qnorm(_floats(el.attrib.get('quat', '1 0 0 0')))
            b = Body(el.attrib['name'], parent, pos, quat, cc)
            bid = len(self.bodies)
            self.bodies.append(b)
            for ch in el:
                if ch.tag in ('joint', 'freejoint'):
                    a = self._attrs(ch, 'joint', cc) if ch.tag == 'joint' else {'name': ch.attrib['name'], 'type': 'free'}
                    if ch.tag == 'freejoint':
                        a['type'] = 'free'
                    b.joints.append(a)
                elif ch.tag == 'geom':
                    b.geoms.append(self._attrs(ch, 'geom', cc))
                elif ch.tag == 'site':
                    b.sites.append(self._attrs(ch, 'site', cc))
            for ch in el:
                if ch.tag == 'body':
                    rec(ch, bid, cc)
        for el in wb:
            if el.tag == 'body':
                rec(el, 0, None)
        self.bname = {b.name: i for i, b in enumerate(self.bodies)}

    # -- rewrites of FruitFly._build / tasks -------------------------------------
    def apply_rewrites(self):
        cfg = self.cfg
        B = self.bodies
        # fruitfly.py:187 removes <freejoint>; base.py:130-134 re-creates it on the
        # attachment frame at the spawn site.  We fuse frame+thorax: thorax carries the
        # free joint with qpos0 = spawn pose.
        thorax = B[self.bname['thorax']]
        thorax.joints = [j for j in thorax.joints if j.get('type') != 'free']
        if not cfg.tethered:
            thorax.joints.insert(0, {'name': 'root', 'type': 'free', '_class': 'main'})
        thorax.pos = np.array(cfg.spawn_pos, float)
        if cfg.ball is not None:
            # BallFloor (tasks/arenas/ball.py:62-69): a sphere on a ball joint.  dm_control puts the arena's bodies before
            # the attached walker; here the ball is the LAST body (its quaternion / dofs are the last entries of qpos / qvel)
            # so that the fly keeps the body and dof numbering of the other tasks.
            pos, radius, density = cfg.ball
            ball = Body('ball', 0, np.array(pos, float), np.array([1., 0, 0, 0]), None)
            ball.joints.append({'name': 'ball', 'type': 'ball', '_class': 'main'})
            ball.geoms.append({'name': 'ball', 'type': 'sphere', 'size': repr(float(radius)), 'density': repr(float(density)), '_class': 'main',
                               'friction': repr(cfg.floor_friction), 'solref': ' '.join(map(repr, cfg.floor_solref)),
                               'solimp': ' '.join(map(repr, cfg.floor_solimp)), 'condim': '3', 'contype': '1', 'conaffinity': '1'})
            B.append(ball); self.bname['ball'] = len(B) - 1

        self.actuators = []
        for el in self.root.find('actuator'):
            a = self._attrs(el, 'actuator', None)
            a['_tag'] = el.tag
            self.actuators.append(a)
        self.tendons = []
        for el in self.root.find('tendon'):
            self.tendons.append({'name': el.attrib['name'],
                                 'joints': [(j.attrib['joint'], float(j.attrib['coef'])) for j in el.findall('joint')]})
        self.excludes = [(e.attrib['body1'], e.attrib['body2']) for e in self.root.find('contact').findall('exclude')]
        self.sensors = [(s.tag, s.attrib['name'], s.attrib['site']) for s in self.root.find('sensor')]
        self.observable_joints = [j['name'] for b in B for j in b.joints if j.get('type') not in ('free', 'ball')]

        def rm_act(name):
            self.actuators = [a for a in self.actuators if a['name'] != name]

        if not cfg.use_legs:
            # fruitfly.py:207-244
            for b in B:
                if _any_in(_NAME_SUBSTR['legs'], b.name):
                    q = np.array([1., 0, 0, 0])
                    hinge = [j for j in b.joints]
                    for j in reversed(hinge):
                        theta = float(j.get('springref', 0))
                        q = qmul(axisangle2q(_floats(j['axis']), theta), q)
                    if hinge:
                        b.quat = qmul(b.quat, q)
            for t in list(self.tendons):
                if _any_in(_NAME_SUBSTR['legs'], t['name']):
                    rm_act(t['name'])
                    self.tendons.remove(t)
            for b in B:
                for j in list(b.joints):
                    if j.get('type') != 'free' and _any_in(_NAME_SUBSTR['legs'], j['name']):
                        rm_act(j['name'])
                        self.observable_joints.remove(j['name'])
                        b.joints.remove(j)
            self.actuators = [a for a in self.actuators
                              if not ('adhere' in a['name'] and _any_in(_NAME_SUBSTR['legs'], a['name']))]
            self.sensors = [s for s in self.sensors if not _any_in(_NAME_SUBSTR['legs'], s[1])]
        for flag, key in ((cfg.use_wings, 'wings'), (cfg.use_mouth, 'mouth'), (cfg.use_antennae, 'antennae')):
            if flag:
                continue
            for b in B:
                for j in b.joints:
                    if j.get('type') != 'free' and _any_in(_NAME_SUBSTR[key], j['name']):
                        rm_act(j['name'])
                        if j['name'] in self.observable_joints:
                            self.observable_joints.remove(j['name'])
            if key == 'mouth':
                self.actuators = [a for a in self.actuators
                                  if not ('adhere' in a['name'] and _any_in(_NAME_SUBSTR['mouth'], a['name']))]
            if key == 'wings':
                self.sensors = [s for s in self.sensors if not _any_in(_NAME_SUBSTR['wings'], s[1])]

        if cfg.use_wings:
            # fruitfly.py:286-306: body pitch and stroke plane
            site = [s for s in thorax.sites if s['name'] == 'hover_up_dir'][0]
            up_dir = qnorm(_floats(site['quat']))   # mjcf keeps the raw attribute; normalise lazily
            up_raw = _floats(site['quat'])
            up_dir_angle = 2 * math.acos(up_raw[0])
            delta = math.radians(cfg.body_pitch_angle) - up_dir_angle
            dq = np.array([math.cos(delta/2), 0, math.sin(delta/2), 0])
            up_raw = qmul(dq, up_raw)
            site['quat'] = ' '.join(repr(float(x)) for x in up_raw)
            spa = math.radians(cfg.stroke_plane_angle)
            spq = np.array([math.cos(spa/2), 0, math.sin(spa/2), 0])
            for quat, wing in ((np.array([0., 0, 0, 1]), 'wing_left'), (np.array([0., -1, 0, 0]), 'wing_right')):
                dq = qmul(qconj(spq), quat)
                new_q = qmul(dq, qconj(up_raw))
                self._change_body_frame(B[self.bname[wing]], new_q)

        # force actuators: fruitfly.py:308-327 -- every `general` actuator loses its affine bias and its own ctrlrange (the
        # single top-level default (-1, 1) applies), gains stay; adhesion actuators are untouched
        if cfg.force_actuators:
            for a in self.actuators:
                if a['_tag'] == 'adhesion':
                    continue
                a.pop('biastype', None); a.pop('biasprm', None)
                a['ctrlrange'] = '-1 1'
        # filters: fruitfly.py:330-340
        dyn = 'filterexact' if cfg.dyntype_filterexact else 'filter'
        for a in self.actuators:
            if a['_tag'] != 'adhesion' and cfg.joint_filter > 0:
                a['dyntype'] = dyn; a['dynprm'] = str(cfg.joint_filter)
            if a['_tag'] == 'adhesion' and cfg.adhesion_filter > 0:
                a['dyntype'] = dyn; a['dynprm'] = str(cfg.adhesion_filter)
        # Flying.__init__: base.py:314-336
        if cfg.wing_gainprm is not None:
            for a in self.actuators:
                if 'wing' in a['name']:
                    a['gainprm'] = str(cfg.wing_gainprm)
        if cfg.wing_stiffness is not None:
            for b in B:
                for j in b.joints:
                    if j.get('type') != 'free' and 'wing' in j['name']:
                        j['stiffness'] = str(cfg.wing_stiffness)
                        j['damping'] = str(cfg.wing_damping)
        if cfg.claw_friction is not None:
            for b in B:
                for g in b.geoms:
                    if g['_class'] == 'adhesion-collision':
                        g['friction'] = str(cfg.claw_friction)
        if cfg.thorax_child_excludes:
            ti = self.bname['thorax']
            for b in B:
                if b.parent == ti:
                    self.excludes.append(('thorax', b.name))
        if cfg.wing_leg_excludes:
            for b in B:
                if _any_in(['coxa', 'femur', 'tibia', 'tarsus', 'claw'], b.name):
                    for w in ('wing_left', 'wing_right'):
                        self.excludes.append((b.name, w))

        # action <-> ctrl maps: fruitfly.py:342-379
        names = [a['name'] for a in self.actuators]
        ctrl_idx = {}
        for cls in _ACTION_CLASSES:
            idx = [i for i, n in enumerate(names) if _any_in(_NAME_SUBSTR[cls], n) and 'adhere' not in n]
            ctrl_idx[cls] = idx
        ctrl_idx['adhesion'] = [i for i, n in enumerate(names) if 'adhere' in n]
        self.action_to_ctrl = []
        for cls in _ACTION_CLASSES:
            self.action_to_ctrl.extend(ctrl_idx[cls])
        self.ctrl_idx = ctrl_idx
        # environment-action indices per class (fruitfly.py:360-379): classes in order, user actions last
        num = {c: len(ctrl_idx[c]) for c in _ACTION_CLASSES}; num['user'] = cfg.num_user_actions
        self.action_idx = {}; counter = 0
        for c in _ACTION_CLASSES:
            self.action_idx[c] = list(range(counter, counter + num[c])); counter += num[c]

    def _change_body_frame(self, body: Body, frame_quat):
        # fruitfly.py:90-117 with frame_pos = body.pos
        body_quat = body.quat.copy()
        dquat = qmul(qconj(frame_quat), body_quat)
        body.quat = np.asarray(frame_quat, float)   # NB un-normalised like the reference; normalised at compile
        for coll, tag in ((body.joints, 'joint'), (body.geoms, 'geom'), (body.sites, 'site')):
            for ch in coll:
                if tag == 'joint':
                    # joints have pos and axis; the reference only moves 'pos' (and 'quat' if present)
                    p = _floats(ch.get('pos', '0 0 0'))
                    pin = qrot(qnorm(body_quat), p)
                    ch['pos'] = ' '.join(repr(float(x)) for x in qrot(qnorm(qconj(frame_quat)), pin))
                    continue
                if 'fromto' in ch:
                    raise NotImplementedError('fromto child in re-framed body')
                cq = _floats(ch.get('quat', '1 0 0 0'))
                ch['quat'] = ' '.join(repr(float(x)) for x in qmul(dquat, cq))
                p = _floats(ch.get('pos', '0 0 0'))
                pin = qrot(qnorm(body_quat), p)
                ch['pos'] = ' '.join(repr(float(x)) for x in qrot(qnorm(qconj(frame_quat)), pin))
        body.quat = qnorm(body.quat)

    # -- geom helpers -----------------------------------------------------------
    def _geom_frame(self, g) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        size = _floats(g.get('size', '0 0 0'))
        size = np.concatenate([size, np.zeros(3 - len(size))])
        if 'fromto' in g:
            ft = _floats(g['fromto'])
            vec = ft[0:3] - ft[3:6]
            pos = 0.5 * (ft[0:3] + ft[3:6])
            size = np.array([size[0], np.linalg.norm(vec) / 2, 0.0])
            quat = z2quat(vec)
        else:
            pos = _floats(g.get('pos', '0 0 0'))
            if 'euler' in g:
                quat = euler2q(_floats(g['euler']))
            else:
                quat = qnorm(_floats(g.get('quat', '1 0 0 0')))
        return pos, quat, size

    def _geom_mass_props(self, g, gtype, size):
        """-> (mass, com_local[3], inertia_local[3,3]) in the geom frame (before geom pos/quat)."""
        if gtype == 7:
            fn = self.meshfile[g['mesh']]
            path = os.path.join(self.asset_dir, fn)
            if not os.path.exists(path):
                return None
            V, F = load_obj(path, self.mesh_scale)
            vol, com, I = mesh_props(V, F)
        elif gtype == GEOM_BOX:
            vol = 8 * size[0] * size[1] * size[2]
            com = np.zeros(3)
            I = vol / 3 * np.diag([size[1]**2 + size[2]**2, size[0]**2 + size[2]**2, size[0]**2 + size[1]**2])
        elif gtype == GEOM_ELLIPSOID:
            vol = 4 / 3 * math.pi * size[0] * size[1] * size[2]
            com = np.zeros(3)
            I = vol / 5 * np.diag([size[1]**2 + size[2]**2, size[0]**2 + size[2]**2, size[0]**2 + size[1]**2])
        elif gtype == GEOM_SPHERE:
            vol = 4 / 3 * math.pi * size[0]**3
            com = np.zeros(3)
            I = 0.4 * vol * size[0]**2 * np.eye(3)
        elif gtype == GEOM_CAPSULE:
            r, h = size[0], size[1] * 2
            vc = math.pi * r * r * h
            vs = 4 / 3 * math.pi * r**3
            vol = vc + vs
            com = np.zeros(3)
            izz = vc * r * r / 2 + vs * 2 * r * r / 5
            ixx = vc * (h * h / 12 + r * r / 4) + vs * (2 * r * r / 5 + h * h / 4 + 3 * h * r / 8)
            I = np.diag([ixx, ixx, izz])
        elif gtype == GEOM_CYLINDER:
            r, h = size[0], size[1] * 2
            vol = math.pi * r * r * h
            com = np.zeros(3)
            I = vol * np.diag([(3 * r * r + h * h) / 12, (3 * r * r + h * h) / 12, r * r / 2])
        else:
            raise ValueError(gtype)
        if 'mass' in g:
            mass = float(g['mass'])
            scale = mass / vol if vol > 0 else 0.0
        else:
            scale = float(g.get('density', 1000.0))
            mass = scale * vol
        return mass, com, I * scale

    # -- compile -----------------------------------------------------------------
    def _actuator_arrays(self, tname=None):
        """Per-actuator tables in MuJoCo's conventions.  `tname` (tendon name -> id) is only known to the full compile."""
        act = dict(trntype=[], trnid=[], dyntype=[], dynprm=[], gainprm=[], biastype=[], biasprm=[],
                   ctrlrange=[], ctrllimited=[], forcerange=[], forcelimited=[], name=[])
        for a in self.actuators:
            act['name'].append(a['name'])
            if a['_tag'] == 'adhesion':
                act['trntype'].append(TRN_BODY); act['trnid'].append(self.bname[a['body']])
                act['biastype'].append(BIAS_NONE); act['biasprm'].append(np.zeros(3))
            else:
                if 'joint' in a:
                    act['trntype'].append(TRN_JOINT); act['trnid'].append(self.jname[a['joint']] if hasattr(self, 'jname') and a['joint'] in getattr(self, 'jname', {}) else -1)
                else:
                    act['trntype'].append(TRN_TENDON); act['trnid'].append(tname[a['tendon']] if tname else -1)
                bt = a.get('biastype', 'none')
                act['biastype'].append(BIAS_AFFINE if bt == 'affine' else BIAS_NONE)
                bp = _floats(a.get('biasprm', '0 0 0'))
                act['biasprm'].append(np.concatenate([bp, np.zeros(3 - len(bp))]))
            dt = a.get('dyntype', 'none')
            act['dyntype'].append({'none': DYN_NONE, 'filter': DYN_FILTER, 'filterexact': DYN_FILTEREXACT}[dt])
            act['dynprm'].append(float(a.get('dynprm', '1').split()[0]))
            gp = _floats(a.get('gainprm', '1 0 0'))
            act['gainprm'].append(np.concatenate([gp, np.zeros(3 - len(gp))]))
            cr = _floats(a.get('ctrlrange', '0 0'))
            act['ctrlrange'].append(cr)
            cl = a.get('ctrllimited', 'auto')
            act['ctrllimited'].append(int((cl == 'true' or cl == 'auto') and 'ctrlrange' in a))
            fr = _floats(a.get('forcerange', '0 0'))
            act['forcerange'].append(fr)
            fl = a.get('forcelimited', 'auto')
            act['forcelimited'].append(int((fl == 'true' or fl == 'auto') and 'forcerange' in a))
        return act

    def actuator_spec(self) -> Dict[str, object]:
        """The part of FruitFly._build the acceptance sweep of the reference checks (tests/test_flywalker.py:36-168):
        actuator names / types / dynamics / gains / ranges, `_ctrl_indices`, `_action_indices`, the action -> ctrl map and
        the action spec -- without the (seconds-long) inertia / contact-pair compile."""
        self.parse_bodies(); self.apply_rewrites()
        act = self._actuator_arrays(None)
        out = {k: np.array(v) for k, v in act.items()}
        out['ctrl_indices'] = {c: (self.ctrl_idx[c] or None) for c in _ACTION_CLASSES}
        out['action_indices'] = dict(self.action_idx)
        out['action_to_ctrl'] = np.array(self.action_to_ctrl, int)
        names = [act['name'][i] for i in self.action_to_ctrl] + [f'user_{k}' for k in range(self.cfg.num_user_actions)]
        rng = [tuple(act['ctrlrange'][i]) for i in self.action_to_ctrl] + [(-1.0, 1.0)]*self.cfg.num_user_actions
        out['action_names'] = names; out['action_minimum'] = np.array([r[0] for r in rng]); out['action_maximum'] = np.array([r[1] for r in rng])
        return out

    def compile(self) -> Dict[str, np.ndarray]:
        cfg = self.cfg
        self.parse_bodies()
        self.apply_rewrites()
        B = self.bodies
        nbody = len(B)
        m: Dict[str, np.ndarray] = {}

        # ---- joints / dofs
        jnt = dict(type=[], qposadr=[], dofadr=[], bodyid=[], pos=[], axis=[], stiffness=[], range=[],
                   limited=[], qpos_spring=[], qpos0=[], solref=[], solimp=[], margin=[], name=[],
                   damping=[], armature=[], springdamper=[])
        body_jntadr = np.zeros(nbody, int); body_jntnum = np.zeros(nbody, int)
        body_dofadr = np.zeros(nbody, int); body_dofnum = np.zeros(nbody, int)
        nq = nv = 0
        dof_body, dof_jnt = [], []
        for bi, b in enumerate(B):
            body_jntadr[bi] = len(jnt['type']); body_dofadr[bi] = nv
            for j in b.joints:
                free = j.get('type') == 'free'; ballj = j.get('type') == 'ball'
                jnt['type'].append(JNT_FREE if free else (JNT_BALL if ballj else JNT_HINGE))
                jnt['qposadr'].append(nq); jnt['dofadr'].append(nv); jnt['bodyid'].append(bi)
                jnt['name'].append(j['name'])
                jnt['pos'].append(_floats(j.get('pos', '0 0 0')))
                ax = _floats(j.get('axis', '0 0 1'))
                jnt['axis'].append(ax / np.linalg.norm(ax))
                jnt['stiffness'].append(float(j.get('stiffness', 0)))
                jnt['damping'].append(0.0 if (free or ballj) else float(j.get('damping', 0)))
                jnt['armature'].append(0.0 if (free or ballj) else float(j.get('armature', 0)))
                sd = _floats(j.get('springdamper', '0 0'))
                jnt['springdamper'].append(sd)
                rng = _floats(j.get('range', '0 0'))
                jnt['range'].append(rng)
                lim = j.get('limited', 'auto')
                limited = (lim == 'true') or (lim == 'auto' and 'range' in j)
                if lim == 'true' and 'range' not in j:
                    limited = False if free else True
                jnt['limited'].append(0 if (free or ballj) else int(limited and 'range' in j))
                jnt['solref'].append(_floats(j.get('solreflimit', '0.02 1')))
                si = _floats(j.get('solimplimit', '0.9 0.95 0.001 0.5 2'))
                jnt['solimp'].append(np.concatenate([si, [0.9, 0.95, 0.001, 0.5, 2][len(si):]]))
                jnt['margin'].append(float(j.get('margin', 0)))
                if free:
                    jnt['qpos0'].append(np.concatenate([b.pos, b.quat]))
                    jnt['qpos_spring'].append(np.concatenate([b.pos, b.quat]))
                    for _ in range(6):
                        dof_body.append(bi); dof_jnt.append(len(jnt['type']) - 1)
                    nq += 7; nv += 6
                elif ballj:
                    jnt['qpos0'].append(np.array([1., 0, 0, 0])); jnt['qpos_spring'].append(np.array([1., 0, 0, 0]))
                    for _ in range(3):
                        dof_body.append(bi); dof_jnt.append(len(jnt['type']) - 1)
                    nq += 4; nv += 3
                else:
                    jnt['qpos0'].append(np.array([float(j.get('ref', 0))]))
                    jnt['qpos_spring'].append(np.array([float(j.get('springref', 0))]))
                    dof_body.append(bi); dof_jnt.append(len(jnt['type']) - 1)
                    nq += 1; nv += 1
            body_jntnum[bi] = len(jnt['type']) - body_jntadr[bi]
            body_dofnum[bi] = nv - body_dofadr[bi]
        njnt = len(jnt['type'])
        self.jname = {n: i for i, n in enumerate(jnt['name'])}
        dof_body = np.array(dof_body); dof_jnt = np.array(dof_jnt)
        # dof parent (previous dof in the same body, else last dof of nearest ancestor with dofs)
        dof_parent = -np.ones(nv, int)
        for d in range(nv):
            bi = dof_body[d]
            if d > body_dofadr[bi]:
                dof_parent[d] = d - 1
            else:
                p = B[bi].parent
                while p > 0 and body_dofnum[p] == 0:
                    p = B[p].parent
                dof_parent[d] = body_dofadr[p] + body_dofnum[p] - 1 if p > 0 and body_dofnum[p] > 0 else -1

        qpos0 = np.concatenate(jnt['qpos0']); qpos_spring = np.concatenate(jnt['qpos_spring'])

        # ---- geoms: mass properties per body + collision geoms
        body_mass = np.zeros(nbody); body_ipos = np.zeros((nbody, 3)); body_iquat = np.tile([1., 0, 0, 0], (nbody, 1))
        body_inertia = np.zeros((nbody, 3))
        cgeoms = []   # colliding + fluid geoms kept for the runtime
        pending_missing = {}
        for bi, b in enumerate(B):
            parts = []
            for g in b.geoms:
                gtype = _GEOM_TYPES[g.get('type', 'sphere')]
                pos, quat, size = self._geom_frame(g)
                contype = int(g.get('contype', 1)); conaff = int(g.get('conaffinity', 1))
                is_fluid = 'fluid' in g.get('name', '')
                mp = self._geom_mass_props(g, gtype, size)
                if mp is None:
                    self.missing_meshes.append(g['mesh'])
                    if 'mass' in g and float(g['mass']) == 0.0:
                        continue
                    pending_missing.setdefault(bi, []).append(g)
                    continue
                mass, com, I = mp
                if mass > 0:
                    R = q2mat(quat)
                    parts.append((mass, pos + R @ com, R @ I @ R.T))
                if gtype != 7 and (contype or conaff or is_fluid):
                    cgeoms.append(dict(name=g['name'], type=gtype, body=bi, pos=pos, quat=quat, size=size,
                                       contype=contype, conaffinity=conaff, condim=int(g.get('condim', 3)),
                                       friction=np.concatenate([_floats(g.get('friction', '1 0.005 0.0001')),
                                                                [1, 0.005, 0.0001][len(_floats(g.get('friction', '1 0.005 0.0001'))):]]),
                                       margin=float(g.get('margin', 0)), gap=float(g.get('gap', 0)),
                                       solref=_floats(g.get('solref', '0.02 1')),
                                       solimp=np.concatenate([_floats(g.get('solimp', '0.9 0.95 0.001')), [0.5, 2.0]])[:5],
                                       fluid=is_fluid))
            b._parts = parts
        # missing-mesh bodies (thorax, head): approximate by their collision ellipsoids
        for bi, gl in pending_missing.items():
            b = B[bi]
            ells = [c for c in cgeoms if c['body'] == bi and c['type'] == GEOM_ELLIPSOID and (c['contype'] or c['conaffinity'])]
            if b.name == 'thorax':
                total = sum(float(g['mass']) for g in gl if 'mass' in g)
            elif b.name == 'head':
                total = None   # resolved after subtree masses are known
            else:
                raise RuntimeError(f'missing mesh on unexpected body {b.name}')
            b._missing = (total, ells)
            self.notes.append(f'body {b.name}: mesh missing; inertia from collision ellipsoids')

        def finish_body(bi):
            b = B[bi]
            parts = list(b._parts)
            if hasattr(b, '_missing'):
                total, ells = b._missing
                vols = [4 / 3 * math.pi * np.prod(c['size']) for c in ells]
                for c, v in zip(ells, vols):
                    mass = total * v / sum(vols)
                    s = c['size']
                    I = mass / 5 * np.diag([s[1]**2 + s[2]**2, s[0]**2 + s[2]**2, s[0]**2 + s[1]**2])
                    R = q2mat(c['quat'])
                    parts.append((mass, c['pos'], R @ I @ R.T))
            mass = sum(p[0] for p in parts)
            if mass <= 0:
                return
            com = sum(p[0] * p[1] for p in parts) / mass
            I = np.zeros((3, 3))
            for pm, pc, pI in parts:
                d = pc - com
                I += pI + pm * (d @ d * np.eye(3) - np.outer(d, d))
            w, V = np.linalg.eigh(I)
            order = np.argsort(-w)
            w = w[order]; V = V[:, order]
            if np.linalg.det(V) < 0:
                V[:, 2] *= -1
            body_mass[bi] = mass; body_ipos[bi] = com; body_iquat[bi] = mat2q(V); body_inertia[bi] = w

        for bi in range(1, nbody):
            if not (hasattr(B[bi], '_missing') and B[bi]._missing[0] is None):
                finish_body(bi)
        # head mass from the pinned subtree mass
        if 'head' in self.bname and hasattr(B[self.bname['head']], '_missing'):
            hi = self.bname['head']
            desc = [i for i in range(nbody) if self._is_descendant(i, hi) and i != hi]
            total = HEAD_SUBTREE_MASS_PIN - body_mass[desc].sum()
            B[hi]._missing = (total, B[hi]._missing[1])
            finish_body(hi)
        subtreemass = body_mass.copy()
        for bi in range(nbody - 1, 0, -1):
            subtreemass[B[bi].parent] += subtreemass[bi]

        # ---- floor (dm_control floors.Floor(): plane at z=0, default friction then task override)
        geoms = []
        if cfg.floor:
            geoms.append(dict(name='floor', type=GEOM_PLANE, body=0, pos=np.zeros(3), quat=np.array([1., 0, 0, 0]),
                              size=np.array([8., 8, 0.25]), contype=1 if cfg.floor_contacts else 0,
                              conaffinity=1 if cfg.floor_contacts else 0, condim=3,
                              friction=np.array([cfg.floor_friction, 0.005, 0.0001]), margin=0.0, gap=0.0,
                              solref=np.array(cfg.floor_solref), solimp=np.array(list(cfg.floor_solimp) + [0.5, 2.0]),
                              fluid=False))
        geoms.extend(cgeoms)
        ngeom = len(geoms)

        # ---- sites
        sites = []
        for bi, b in enumerate(B):
            for s in b.sites:
                pos, quat, size = self._geom_frame(s)
                sites.append(dict(name=s['name'], body=bi, pos=pos, quat=quat, size=size,
                                  type=_GEOM_TYPES[s.get('type', 'sphere')]))
        self.sname = {s['name']: i for i, s in enumerate(sites)}

        # ---- tendons (fixed)
        ten_adr, ten_num, wrap_dof, wrap_coef = [], [], [], []
        for t in self.tendons:
            ten_adr.append(len(wrap_dof)); ten_num.append(len(t['joints']))
            for jn, c in t['joints']:
                wrap_dof.append(jnt['dofadr'][self.jname[jn]]); wrap_coef.append(c)
        tname = {t['name']: i for i, t in enumerate(self.tendons)}

        # ---- actuators
        nu = len(self.actuators)
        act = self._actuator_arrays(tname)
        act_adr = -np.ones(nu, int)
        na = 0
        for i in range(nu):
            if act['dyntype'][i] != DYN_NONE:
                act_adr[i] = na; na += 1

        # ---- store basics
        depth = np.zeros(nbody, int)
        for bi in range(1, nbody):
            depth[bi] = depth[B[bi].parent] + 1
        m['body_parent'] = np.array([b.parent for b in B])
        m['body_depth'] = depth
        m['body_pos'] = np.array([b.pos for b in B]); m['body_quat'] = np.array([qnorm(b.quat) for b in B])
        m['body_ipos'] = body_ipos; m['body_iquat'] = body_iquat
        m['body_mass'] = body_mass; m['body_inertia'] = body_inertia; m['body_subtreemass'] = subtreemass
        m['body_jntadr'] = body_jntadr; m['body_jntnum'] = body_jntnum
        m['body_dofadr'] = body_dofadr; m['body_dofnum'] = body_dofnum
        m['jnt_type'] = np.array(jnt['type']); m['jnt_qposadr'] = np.array(jnt['qposadr'])
        m['jnt_dofadr'] = np.array(jnt['dofadr']); m['jnt_bodyid'] = np.array(jnt['bodyid'])
        m['jnt_pos'] = np.array(jnt['pos']); m['jnt_axis'] = np.array(jnt['axis'])
        m['jnt_stiffness'] = np.array(jnt['stiffness']); m['jnt_range'] = np.array(jnt['range'])
        m['jnt_limited'] = np.array(jnt['limited']); m['jnt_solref'] = np.array(jnt['solref'])
        m['jnt_solimp'] = np.array(jnt['solimp']); m['jnt_margin'] = np.array(jnt['margin'])
        m['qpos0'] = qpos0; m['qpos_spring'] = qpos_spring
        m['dof_bodyid'] = dof_body; m['dof_jntid'] = dof_jnt; m['dof_parentid'] = dof_parent
        m['dof_armature'] = np.array([jnt['armature'][j] for j in dof_jnt])
        m['dof_damping'] = np.array([jnt['damping'][j] for j in dof_jnt])
        for k in ('type', 'body', 'contype', 'conaffinity', 'condim'):
            m['geom_' + ('bodyid' if k == 'body' else k)] = np.array([g[k] for g in geoms])
        for k in ('pos', 'quat', 'size', 'friction', 'solref', 'solimp'):
            m['geom_' + k] = np.array([g[k] for g in geoms], float)
        m['geom_margin'] = np.array([g['margin'] for g in geoms]); m['geom_gap'] = np.array([g['gap'] for g in geoms])
        m['geom_isfluid'] = np.array([int(g['fluid']) for g in geoms])
        m['site_bodyid'] = np.array([s['body'] for s in sites]); m['site_pos'] = np.array([s['pos'] for s in sites])
        m['site_quat'] = np.array([s['quat'] for s in sites]); m['site_size'] = np.array([s['size'] for s in sites])
        m['site_type'] = np.array([s['type'] for s in sites])
        m['tendon_adr'] = np.array(ten_adr, int); m['tendon_num'] = np.array(ten_num, int)
        m['wrap_dofid'] = np.array(wrap_dof, int); m['wrap_coef'] = np.array(wrap_coef, float)
        m['actuator_trntype'] = np.array(act['trntype']); m['actuator_trnid'] = np.array(act['trnid'])
        m['actuator_dyntype'] = np.array(act['dyntype']); m['actuator_dynprm'] = np.array(act['dynprm'])
        m['actuator_gainprm'] = np.array(act['gainprm']); m['actuator_biastype'] = np.array(act['biastype'])
        m['actuator_biasprm'] = np.array(act['biasprm']); m['actuator_ctrlrange'] = np.array(act['ctrlrange'])
        m['actuator_ctrllimited'] = np.array(act['ctrllimited']); m['actuator_forcerange'] = np.array(act['forcerange'])
        m['actuator_forcelimited'] = np.array(act['forcelimited']); m['actuator_actadr'] = act_adr
        m['action_to_ctrl'] = np.array(self.action_to_ctrl, int)
        opt = self.root.find('option').attrib
        m['opt_timestep'] = np.array(cfg.physics_timestep)
        m['opt_control_timestep'] = np.array(cfg.control_timestep)
        m['opt_gravity'] = _floats(opt.get('gravity', '0 0 -9.81'))
        m['opt_density'] = np.array(float(opt.get('density', 0))); m['opt_viscosity'] = np.array(float(opt.get('viscosity', 0)))
        m['opt_noslip_iterations'] = np.array(int(opt.get('noslip_iterations', 0)))
        m['opt_impratio'] = np.array(float(opt.get('impratio', 1)))
        m['opt_cone_elliptic'] = np.array(int(opt.get('cone', 'pyramidal') == 'elliptic'))
        m['opt_iterations'] = np.array(int(opt.get('iterations', 100)))
        m['opt_tolerance'] = np.array(float(opt.get('tolerance', 1e-8)))
        m['opt_noslip_tolerance'] = np.array(float(opt.get('noslip_tolerance', 1e-6)))

        # names (for tests / action spec / obs)
        m['names_body'] = np.array([b.name for b in B]); m['names_jnt'] = np.array(jnt['name'])
        m['names_geom'] = np.array([g['name'] for g in geoms]); m['names_site'] = np.array([s['name'] for s in sites])
        m['names_actuator'] = np.array(act['name']); m['names_tendon'] = np.array([t['name'] for t in self.tendons])
        m['observable_joints'] = np.array([self.jname[n] for n in self.observable_joints], int)
        # leg joints (flight with enabled legs resets them to / rewards them at their spring reference: flight_imitation.py:142-144,196-203)
        m['leg_joints'] = np.array([i for i, n in enumerate(jnt['name']) if _any_in(_NAME_SUBSTR['legs'], n) and jnt['type'][i] == JNT_HINGE], int)

        # ---- reference-configuration quantities (M0, invweight0, springdamper)
        self._set0(m, jnt)
        # ---- collision pairs
        self._pairs(m, geoms)
        # ---- fluid ellipsoid coefficients
        self._fluid(m, geoms)
        # ---- sensors & observation helper ids
        self._sensors(m, sites)
        # ---- derived index tables used by both engines
        nv_ = len(m['dof_bodyid'])
        madr = np.zeros(nv_ + 1, int)
        for d in range(nv_):
            cnt = 0; k = d
            while k >= 0:
                cnt += 1; k = m['dof_parentid'][k]
            madr[d + 1] = madr[d] + cnt
        m['dof_Madr'] = madr
        m['stat_meaninertia'] = np.array(float(np.mean(m['dof_M0'])))
        rootid = np.zeros(nbody, int)
        for bi in range(1, nbody):
            rootid[bi] = bi if m['body_parent'][bi] == 0 else rootid[m['body_parent'][bi]]
        m['body_rootid'] = rootid
        # task-level index tables (flight_imitation.py:60-63): positions of the wing / user entries in the action
        names_act = [a['name'] for a in self.actuators]
        act_order = [names_act[i] for i in self.action_to_ctrl]
        m['wing_action_idx'] = np.array([k for k, n in enumerate(act_order) if 'wing' in n], int)
        m['user_action_idx'] = np.array(len(act_order) if cfg.num_user_actions else -1)
        m['task_id'] = np.array({'walk_imitation': 0, 'flight_imitation': 1, 'walk_on_ball': 2}[cfg.name])
        m['com_offset'] = np.array([-0.03697732, 0.00029205, -0.0142447])     # tasks/task_utils.py:237
        m['notes'] = np.array(self.notes)
        m['config_name'] = np.array(cfg.name)
        m['num_user_actions'] = np.array(cfg.num_user_actions)
        return m

    def _is_descendant(self, i, anc):
        while i > 0:
            if i == anc:
                return True
            i = self.bodies[i].parent
        return False

    # -- kinematics at qpos0 in numpy (compile-time only) ---------------------------
    def _fk0(self, m):
        nbody = len(self.bodies)
        xpos = np.zeros((nbody, 3)); xquat = np.tile([1., 0, 0, 0], (nbody, 1))
        for bi in range(1, nbody):
            p = m['body_parent'][bi]
            if m['body_jntnum'][bi] and m['jnt_type'][m['body_jntadr'][bi]] == JNT_FREE:
                q0 = m['qpos0'][m['jnt_qposadr'][m['body_jntadr'][bi]]:][:7]
                xpos[bi] = q0[:3]; xquat[bi] = qnorm(q0[3:])
            else:
                xpos[bi] = xpos[p] + qrot(xquat[p], m['body_pos'][bi])
                xquat[bi] = qmul(xquat[p], m['body_quat'][bi])
            # hinge joints at ref: no rotation
        return xpos, xquat

    def _set0(self, m, jnt):
        nbody = len(self.bodies); nv = len(m['dof_bodyid'])
        xpos, xquat = self._fk0(m)
        xipos = np.array([xpos[b] + qrot(xquat[b], m['body_ipos'][b]) for b in range(nbody)])
        # dof axes in world
        dof_axis = np.zeros((nv, 3)); dof_anchor = np.zeros((nv, 3)); dof_isrot = np.zeros(nv, bool); dof_istrans = np.zeros(nv, bool)
        for j in range(len(m['jnt_type'])):
            b = m['jnt_bodyid'][j]; d = m['jnt_dofadr'][j]
            if m['jnt_type'][j] == JNT_FREE:
                R = q2mat(xquat[b])
                for k in range(3):
                    dof_axis[d + k] = np.eye(3)[k]; dof_istrans[d + k] = True
                    dof_axis[d + 3 + k] = R[:, k]; dof_isrot[d + 3 + k] = True; dof_anchor[d + 3 + k] = xpos[b]
            elif m['jnt_type'][j] == JNT_BALL:
                R = q2mat(xquat[b])
                for k in range(3):
                    dof_axis[d + k] = R[:, k]; dof_isrot[d + k] = True; dof_anchor[d + k] = xpos[b] + qrot(xquat[b], m['jnt_pos'][j])
            else:
                dof_axis[d] = qrot(xquat[b], m['jnt_axis'][j]); dof_isrot[d] = True
                dof_anchor[d] = xpos[b] + qrot(xquat[b], m['jnt_pos'][j])
        anc = [[] for _ in range(nbody)]   # dofs affecting each body
        for b in range(1, nbody):
            p = m['body_parent'][b]
            anc[b] = anc[p] + list(range(m['body_dofadr'][b], m['body_dofadr'][b] + m['body_dofnum'][b]))

        def jac(b, point):
            Jp = np.zeros((3, nv)); Jr = np.zeros((3, nv))
            for d in anc[b]:
                if dof_istrans[d]:
                    Jp[:, d] = dof_axis[d]
                else:
                    Jr[:, d] = dof_axis[d]; Jp[:, d] = np.cross(dof_axis[d], point - dof_anchor[d])
            return Jp, Jr
        M = np.diag(m['dof_armature'].astype(float))
        for b in range(1, nbody):
            if m['body_mass'][b] <= 0:
                continue
            Jp, Jr = jac(b, xipos[b])
            R = q2mat(qmul(xquat[b], m['body_iquat'][b]))
            Iw = R @ np.diag(m['body_inertia'][b]) @ R.T
            M += m['body_mass'][b] * Jp.T @ Jp + Jr.T @ Iw @ Jr
        m['dof_M0'] = np.diag(M).copy()
        # springdamper (haltere): stiffness/damping from joint inertia at qpos0
        for j in range(len(m['jnt_type'])):
            sd = jnt['springdamper'][j]
            if sd[0] > 0 and sd[1] > 0:
                d = m['jnt_dofadr'][j]
                inertia = M[d, d]
                m['jnt_stiffness'][j] = inertia / max(MINVAL, sd[0]**2 * sd[1]**2)
                m['dof_damping'][d] = 2 * inertia / max(MINVAL, sd[0])
        Minv = np.linalg.inv(M)
        biw = np.zeros((nbody, 2))
        for b in range(1, nbody):
            Jp, Jr = jac(b, xipos[b])
            biw[b, 0] = np.trace(Jp @ Minv @ Jp.T) / 3
            biw[b, 1] = np.trace(Jr @ Minv @ Jr.T) / 3
        m['body_invweight0'] = biw
        diw = np.zeros(nv)
        for j in range(len(m['jnt_type'])):
            d = m['jnt_dofadr'][j]
            if m['jnt_type'][j] == JNT_FREE:
                diw[d:d + 3] = np.mean(np.diag(Minv)[d:d + 3]); diw[d + 3:d + 6] = np.mean(np.diag(Minv)[d + 3:d + 6])
            elif m['jnt_type'][j] == JNT_BALL:
                diw[d:d + 3] = np.mean(np.diag(Minv)[d:d + 3])
            else:
                diw[d] = Minv[d, d]
        m['dof_invweight0'] = diw
        tiw = np.zeros(len(m['tendon_adr']))
        for t in range(len(tiw)):
            J = np.zeros(nv)
            for w in range(m['tendon_adr'][t], m['tendon_adr'][t] + m['tendon_num'][t]):
                J[m['wrap_dofid'][w]] = m['wrap_coef'][w]
            tiw[t] = J @ Minv @ J
        m['tendon_invweight0'] = tiw
        m['M0_full'] = M

    def _pairs(self, m, geoms):
        B = self.bodies
        nbody = len(B)
        # weld ids: bodies without dofs are welded to their parent
        weld = np.arange(nbody)
        for b in range(1, nbody):
            if m['body_dofnum'][b] == 0:
                weld[b] = weld[m['body_parent'][b]]
        excl = set()
        for a, b in self.excludes:
            ia, ib = self.bname[a], self.bname[b]
            excl.add((min(ia, ib), max(ia, ib)))
        pairs = []
        ng = len(geoms)
        for i in range(ng):
            for j in range(i + 1, ng):
                gi, gj = geoms[i], geoms[j]
                if not ((gi['contype'] & gj['conaffinity']) or (gj['contype'] & gi['conaffinity'])):
                    continue
                b1, b2 = gi['body'], gj['body']
                w1, w2 = weld[b1], weld[b2]
                if w1 == w2:
                    continue
                # parent-child filter (not applied when the parent is the world)
                p1, p2 = weld[m['body_parent'][w1]] if w1 > 0 else -1, weld[m['body_parent'][w2]] if w2 > 0 else -1
                if (w1 != 0 and w2 != 0) and (p1 == w2 or p2 == w1):
                    continue
                if (min(b1, b2), max(b1, b2)) in excl:
                    continue
                if gi['type'] == GEOM_PLANE and gj['type'] == GEOM_PLANE:
                    continue
                # order so that the lower geom type comes first (collision-function table is upper-triangular)
                a, b = (i, j) if gi['type'] <= gj['type'] else (j, i)
                ga, gb = geoms[a], geoms[b]
                condim = max(ga['condim'], gb['condim'])
                friction = np.maximum(ga['friction'], gb['friction'])
                # solmix = 1 for both -> equal-weight average
                solref = 0.5 * (ga['solref'] + gb['solref'])
                solimp = 0.5 * (ga['solimp'] + gb['solimp'])
                margin = max(ga['margin'], gb['margin']); gap = max(ga['gap'], gb['gap'])
                pairs.append((a, b, condim, friction, solref, solimp, margin, gap))
        m['pair_geom1'] = np.array([p[0] for p in pairs], int); m['pair_geom2'] = np.array([p[1] for p in pairs], int)
        m['pair_condim'] = np.array([p[2] for p in pairs], int)
        # friction laid out as MuJoCo contact.friction[5]: tangent1, tangent2, spin, roll1, roll2
        m['pair_friction'] = np.array([[p[3][0], p[3][0], p[3][1], p[3][2], p[3][2]] for p in pairs], float)
        m['pair_solref'] = np.array([p[4] for p in pairs], float); m['pair_solimp'] = np.array([p[5] for p in pairs], float)
        m['pair_margin'] = np.array([p[6] for p in pairs], float); m['pair_gap'] = np.array([p[7] for p in pairs], float)
        # bounding radius per geom
        rb = np.zeros(len(geoms))
        for i, g in enumerate(geoms):
            s = g['size']
            rb[i] = {GEOM_PLANE: 0.0, GEOM_SPHERE: s[0], GEOM_CAPSULE: s[0] + s[1], GEOM_ELLIPSOID: max(s),
                     GEOM_CYLINDER: math.hypot(s[0], s[1]), GEOM_BOX: np.linalg.norm(s)}[g['type']]
        m['geom_rbound'] = rb

    def _fluid(self, m, geoms):
        """Ellipsoid fluid interaction coefficients (MuJoCo geom_fluid[12] layout, cf.
        flybody/ellipsoid_fluid_model.py:229-237): [enable, blunt, slender, ang, kutta, magnus,
        virtual_mass[3], virtual_inertia[3]]."""
        ng = len(geoms)
        gf = np.zeros((ng, 12))
        if self.cfg.fluidcoef is not None:
            for i, g in enumerate(geoms):
                if not g['fluid']:
                    continue
                gf[i, 0] = 1.0
                gf[i, 1:6] = self.cfg.fluidcoef
                vm, vi = ellipsoid_virtual_inertia(g['size'])
                gf[i, 6:9] = vm; gf[i, 9:12] = vi
        m['geom_fluid'] = gf

    def _sensors(self, m, sites):
        sn = self.sname
        m['sensor_site_thorax'] = np.array(sn['thorax'])
        force_sites = [sn[s[2]] for s in self.sensors if s[0] == 'force']
        touch_sites = [sn[s[2]] for s in self.sensors if s[0] == 'touch']
        m['sensor_force_sites'] = np.array(force_sites, int); m['sensor_touch_sites'] = np.array(touch_sites, int)
        # appendages: end effectors (claw sites) + head site  (fruitfly.py:478-497)
        # (enabled by Walking, base.py:425-428; Flying enables them only with legs, base.py:360-364)
        app = [sn[n] for n in ('claw_T1_left', 'claw_T1_right', 'claw_T2_left', 'claw_T2_right',
                               'claw_T3_left', 'claw_T3_right') if n in sn and self.cfg.use_legs]
        if self.cfg.use_legs:
            app.append(sn['head'])
        m['appendage_sites'] = np.array(app, int)
        # wing joints (qpos addresses) and their springrefs
        wj = [j for j, n in enumerate(m['names_jnt']) if _any_in(['yaw', 'roll', 'pitch'], str(n))]
        m['wing_jnt'] = np.array(wj, int)


def _rj_integrals(a, b, c):
    """kappa-type elliptic integrals for the added-mass of a tri-axial ellipsoid (Lamb §114):
    alpha0 = abc * int_0^inf du / ((a^2+u) * Delta), Delta = sqrt((a^2+u)(b^2+u)(c^2+u)); likewise beta0, gamma0.
    Evaluated with Gauss-Legendre quadrature after the substitution u = t/(1-t) * s."""
    from numpy.polynomial.legendre import leggauss
    x, w = leggauss(400)
    t = 0.5 * (x + 1); w = 0.5 * w
    s = (a * b * c) ** (2.0 / 3.0)
    u = s * t / (1 - t); du = s / (1 - t) ** 2
    delta = np.sqrt((a * a + u) * (b * b + u) * (c * c + u))
    out = []
    for d in (a, b, c):
        out.append(a * b * c * np.sum(w * du / ((d * d + u) * delta)))
    return out


def ellipsoid_virtual_inertia(size):
    """Virtual (added) mass and inertia per unit fluid density of an ellipsoid with semi-axes
    ``size`` (potential-flow result, Lamb, Hydrodynamics §114-115).  These are compile-time
    MuJoCo outputs that the reference cannot derive."""
    a, b, c = [float(x) for x in size]
    al, be, ga = _rj_integrals(a, b, c)
    vol = 4 / 3 * math.pi * a * b * c
    vm = np.array([vol * al / (2 - al), vol * be / (2 - be), vol * ga / (2 - ga)])
    Ifac = vol / 5

    def vin(p, q, ip, iq):
        # rotation about the axis perpendicular to semi-axes p, q with integrals ip, iq
        num = (p * p - q * q) ** 2 * (iq - ip)
        den = 2 * (p * p - q * q) + (p * p + q * q) * (ip - iq)
        return Ifac * num / den if abs(den) > 0 else 0.0
    vi = np.array([vin(b, c, be, ga), vin(c, a, ga, al), vin(a, b, al, be)])
    return vm, vi


# ----------------------------------------------------------------------------
def compile_model(xml_path: str, cfg: TaskConfig) -> Dict[str, np.ndarray]:
    return FlyCompiler(xml_path, cfg).compile()


def save_model(m: Dict[str, np.ndarray], path: str):
    np.savez_compressed(path, **m)
