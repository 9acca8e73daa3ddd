"""Recording fake of the PyBullet client surface the reference touches.

Used ONLY by tests/golden/make_goldens.py, in the build container, to execute
the reference's own Python harness logic (action mapping, obs assembly, rewards,
scene construction) without PyBullet, which is absent here.

The fake has no physics.  Every "world read" (getJointState, getLinkState,
getBasePositionAndOrientation, getBaseVelocity, calculateInverseKinematics,
rayTest) returns values from a programmable `world` dict that the golden
script fills with seeded random numbers; every "world write" is appended to
`log`.  getQuaternionFromEuler / getEulerFromQuaternion follow the formulas in
SURVEY.md App. E (their goldens are therefore circular and are pinned with
analytic identities in the tests instead).
"""
import math
import sys
import types

import numpy as np


# --- pure math -------------------------------------------
def quat_from_euler(rpy):
    r, p, y = [float(v) for v in rpy]
    cr, sr = math.cos(r * 0.5), math.sin(r * 0.5)
    cp, sp = math.cos(p * 0.5), math.sin(p * 0.5)
    cy, sy = math.cos(y * 0.5), math.sin(y * 0.5)
    return (sr * cp * cy - cr * sp * sy,
            cr * sp * cy + sr * cp * sy,
            cr * cp * sy - sr * sp * cy,
            cr * cp * cy + sr * sp * sy)


def euler_from_quat(q):
    x, y, z, w = [float(v) for v in q]
    sarg = -2.0 * (x * z - w * y)
    if sarg <= -0.99999:
        return (0.0, -0.5 * math.pi, 2.0 * math.atan2(x, -y))
    if sarg >= 0.99999:
        return (0.0, 0.5 * math.pi, 2.0 * math.atan2(-x, y))
    sqx, sqy, sqz, sqw = x * x, y * y, z * z, w * w
    return (math.atan2(2.0 * (y * z + w * x), sqw - sqx - sqy + sqz),
            math.asin(sarg),
            math.atan2(2.0 * (x * y + w * z), sqw + sqx - sqy - sqz))


UR5_JOINT_TYPES = None  # filled by make_goldens from the URDF (0 rev, 1 prismatic, 4 fixed)
PANDA_JOINT_TYPES = None


def _plain(v):
    """JSON-able copy of an argument."""
    if isinstance(v, np.ndarray):
        return v.tolist()
    if isinstance(v, (np.floating,)):
        return float(v)
    if isinstance(v, (np.integer,)):
        return int(v)
    if isinstance(v, (list, tuple)):
        return [_plain(x) for x in v]
    if isinstance(v, dict):
        return {str(k): _plain(x) for k, x in v.items()}
    return v


class FakeClient:
    # constants (values as in pybullet; only identity matters)
    GEOM_SPHERE, GEOM_BOX, GEOM_CYLINDER, GEOM_MESH = 2, 3, 4, 5
    GEOM_FORCE_CONCAVE_TRIMESH = 1
    JOINT_REVOLUTE, JOINT_PRISMATIC, JOINT_FIXED, JOINT_GEAR = 0, 1, 4, 6
    POSITION_CONTROL, VELOCITY_CONTROL, TORQUE_CONTROL = 2, 0, 1
    URDF_ENABLE_CACHED_GRAPHICS_SHAPES = 1024
    DIRECT, GUI, SHARED_MEMORY = 2, 1, 3
    ER_NO_SEGMENTATION_MASK = 4
    ER_BULLET_HARDWARE_OPENGL = 131072
    COV_ENABLE_GUI = 1

    def __init__(self, connection_mode=None):
        self.log = []
        self.n_shapes = 0
        self.n_bodies = 0
        self.body_kind = {}      # body id -> 'ur5' | 'panda' | 'multibody' | 'tray'
        self.body_links = {}     # body id -> number of joints
        self.world = {}          # programmable reads
        self.ik_queue = []       # successive calculateInverseKinematics returns

    # ---- recording helpers
    def _rec(self, name, args, kwargs, ret=None):
        self.log.append({'fn': name, 'args': _plain(list(args)), 'kwargs': _plain(kwargs), 'ret': _plain(ret)})
        return ret

    def clear_log(self):
        self.log = []

    # ---- setup / writes
    def createCollisionShape(self, *a, **k):
        self.n_shapes += 1
        if 'fileName' in k:
            k = dict(k, fileName='<envs>/' + '/'.join(k['fileName'].split('/')[-2:]))
        return self._rec('createCollisionShape', a, k, self.n_shapes - 1)

    def createVisualShape(self, *a, **k):
        self.n_shapes += 1
        if 'fileName' in k:
            k = dict(k, fileName='<envs>/' + '/'.join(k['fileName'].split('/')[-2:]))
        return self._rec('createVisualShape', a, k, self.n_shapes - 1)

    def createMultiBody(self, *a, **k):
        b = self.n_bodies
        self.n_bodies += 1
        self.body_kind[b] = 'multibody'
        self.body_links[b] = len(k.get('linkMasses', []))
        return self._rec('createMultiBody', a, k, b)

    def loadURDF(self, fileName, *a, **k):
        b = self.n_bodies
        self.n_bodies += 1
        short = fileName.split('/')[-1]
        if 'ur5e2' in short:
            self.body_kind[b] = 'ur5'
            self.body_links[b] = len(UR5_JOINT_TYPES)
        elif 'panda' in short:
            self.body_kind[b] = 'panda'
            self.body_links[b] = len(PANDA_JOINT_TYPES)
        else:
            self.body_kind[b] = 'tray'
            self.body_links[b] = 0
        return self._rec('loadURDF', ['<envs>/' + '/'.join(fileName.split('/')[-2:])] + list(a), k, b)

    def loadTexture(self, *a, **k):
        return 0

    def _write(name):
        def f(self, *a, **k):
            return self._rec(name, a, k, None)
        f.__name__ = name
        return f

    changeDynamics = _write('changeDynamics')
    changeVisualShape = lambda self, *a, **k: None        # visual only; not logged
    setCollisionFilterGroupMask = _write('setCollisionFilterGroupMask')
    setJointMotorControl2 = _write('setJointMotorControl2')
    setJointMotorControlArray = _write('setJointMotorControlArray')
    setPhysicsEngineParameter = _write('setPhysicsEngineParameter')
    setTimeStep = _write('setTimeStep')
    setGravity = _write('setGravity')
    setAdditionalSearchPath = lambda self, *a, **k: None
    resetDebugVisualizerCamera = lambda self, *a, **k: None
    configureDebugVisualizer = lambda self, *a, **k: None
    changeConstraint = _write('changeConstraint')

    # hard state sets also update the programmable world, so a read after a reset sees the reset value
    def resetJointState(self, body, j, q, *a, **k):
        if 'joint' in self.world:
            self.world['joint'][(body, j)] = float(q)
        return self._rec('resetJointState', (body, j, q) + a, k, None)

    def resetBasePositionAndOrientation(self, body, pos, orn):
        if 'base' in self.world:
            self.world['base'][body] = {'pos': [float(v) for v in pos], 'orn': [float(v) for v in orn],
                                        'lin': [0.0, 0.0, 0.0], 'ang': [0.0, 0.0, 0.0]}
        return self._rec('resetBasePositionAndOrientation', (body, pos, orn), {}, None)

    stepSimulation = _write('stepSimulation')

    def createConstraint(self, *a, **k):
        return self._rec('createConstraint', a, k, 0)

    # ---- reads
    def getNumJoints(self, body):
        return self.body_links[body]

    def getJointInfo(self, body, j):
        kind = self.body_kind[body]
        types = UR5_JOINT_TYPES if kind == 'ur5' else PANDA_JOINT_TYPES
        return (j, b'joint%d' % j, types[j])

    def getJointState(self, body, j):
        q = self.world['joint'][(body, j)]
        return (q, 0.0, (0.0,) * 6, 0.0)

    def getLinkState(self, body, link, computeLinkVelocity=0):
        s = self.world['link'][(body, link)]
        return (tuple(s['pos']), tuple(s['orn']), (0, 0, 0), (0, 0, 0, 1), tuple(s['pos']), tuple(s['orn']),
                tuple(s['lin']), tuple(s['ang']))

    def getBasePositionAndOrientation(self, body):
        s = self.world['base'][body]
        return (tuple(s['pos']), tuple(s['orn']))

    def getBaseVelocity(self, body):
        s = self.world['base'][body]
        return (tuple(s['lin']), tuple(s['ang']))

    def calculateInverseKinematics(self, *a, **k):
        ret = self.ik_queue.pop(0)
        self._rec('calculateInverseKinematics', a, k, None)
        return tuple(ret)

    def rayTest(self, a, b):
        self._rec('rayTest', [a, b], {}, None)
        return self.world['ray']

    def getCameraImage(self, *a, **k):
        raise RuntimeError('rendering is out of scope')

    # ---- math
    def getQuaternionFromEuler(self, rpy):
        return quat_from_euler(rpy)

    def getEulerFromQuaternion(self, q):
        return euler_from_quat(q)

    def computeViewMatrixFromYawPitchRoll(self, *a, **k):
        return [0.0] * 16

    def computeProjectionMatrixFOV(self, *a, **k):
        return [0.0] * 16

    def disconnect(self):
        pass


class StaticWorldClient(FakeClient):
    """A FakeClient whose world answers by itself - no physics: joints stay where resetJointState put them, bodies where they were created /
    reset, links at a fixed pose - plus the read-only calls tools/pybullet_replay.py makes on a real PyBullet (getDynamicsInfo,
    calculateMassMatrix, getContactPoints, getPhysicsEngineParameters, getAPIVersion, the full getJointInfo / getLinkState tuples).
    tests/test_pybullet_golden.py runs the replay tool's PyBullet half against it, end to end, so that the half a human with a real PyBullet
    has to run works on the first try.  Whatever it records is FORMAT, never a pin."""

    def __init__(self, connection_mode=None):
        super().__init__(connection_mode)
        self.world = {'joint': {}, 'link': {}, 'base': {}, 'ray': [(-1, -1, 1.0, (0.0, 0.0, 0.0), (0.0, 0.0, 0.0))]}

    def _n_movable(self, body):
        kind = self.body_kind.get(body)
        types_ = UR5_JOINT_TYPES if kind == 'ur5' else (PANDA_JOINT_TYPES if kind == 'panda' else [0] * self.body_links.get(body, 0))
        return sum(1 for t in types_ if t != self.JOINT_FIXED)

    def createMultiBody(self, *a, **k):
        b = super().createMultiBody(*a, **k)
        pos = k.get('basePosition', a[3] if len(a) > 3 else [0.0, 0.0, 0.0])
        orn = k.get('baseOrientation', a[4] if len(a) > 4 else [0.0, 0.0, 0.0, 1.0])
        self.world['base'][b] = {'pos': [float(v) for v in pos], 'orn': [float(v) for v in orn], 'lin': [0.0] * 3, 'ang': [0.0] * 3}
        return b

    def loadURDF(self, fileName, *a, **k):
        b = super().loadURDF(fileName, *a, **k)
        pos = a[0] if len(a) > 0 else k.get('basePosition', [0.0, 0.0, 0.0])
        orn = a[1] if len(a) > 1 else k.get('baseOrientation', [0.0, 0.0, 0.0, 1.0])
        self.world['base'][b] = {'pos': [float(v) for v in pos], 'orn': [float(v) for v in orn], 'lin': [0.0] * 3, 'ang': [0.0] * 3}
        return b

    def getJointState(self, body, j):
        return (self.world['joint'].get((body, j), 0.0), 0.0, (0.0,) * 6, 0.0)

    def getJointInfo(self, body, j):
        kind = self.body_kind[body]
        types_ = UR5_JOINT_TYPES if kind == 'ur5' else (PANDA_JOINT_TYPES if kind == 'panda' else [self.JOINT_PRISMATIC] * 8)
        # (index, name, type, qIndex, uIndex, flags, damping, friction, lower, upper, maxForce, maxVelocity, linkName, axis, parentFramePos, parentFrameOrn, parentIndex)
        return (j, b'joint%d' % j, types_[j], j, j, 0, 0.0, 0.0, -1.0, 1.0, 100.0, 1.0, b'link%d' % j, (0.0, 0.0, 1.0), (0.0, 0.0, 0.0), (0.0, 0.0, 0.0, 1.0), j - 1)

    def getLinkState(self, body, link, computeLinkVelocity=0):
        s = self.world['link'].get((body, link), {'pos': [0.1, 0.2, 0.3 + 0.01 * link], 'orn': [0.0, 0.0, 0.0, 1.0], 'lin': [0.0] * 3, 'ang': [0.0] * 3})
        return (tuple(s['pos']), tuple(s['orn']), (0.0, 0.0, 0.0), (0.0, 0.0, 0.0, 1.0), tuple(s['pos']), tuple(s['orn']), tuple(s['lin']), tuple(s['ang']))

    def calculateInverseKinematics(self, body, *a, **k):
        self._rec('calculateInverseKinematics', (body,) + a, k, None)
        return tuple([0.0] * self._n_movable(body))

    def getDynamicsInfo(self, body, link):
        # (mass, lateral friction, local inertia diagonal, local inertial pos, local inertial orn, restitution, rolling friction, spinning friction,
        #  contact damping, contact stiffness, body type, collision margin)
        return (1.0, 0.5, (1.0, 1.0, 1.0), (0.0, 0.0, 0.0), (0.0, 0.0, 0.0, 1.0), 0.0, 0.0, 0.0, -1.0, -1.0, 2, 0.001)

    def calculateMassMatrix(self, body, q):
        n = len(q)
        return tuple(tuple(1.0 if i == j else 0.0 for j in range(n)) for i in range(n))

    def getContactPoints(self, *a, **k):
        return ()

    def getPhysicsEngineParameters(self):
        return {'fixedTimeStep': 1.0 / 300.0, 'numSolverIterations': 50, 'numSubSteps': 0, 'useSplitImpulse': 0, 'contactBreakingThreshold': 0.02}

    def getAPIVersion(self):
        return 0


class _Box:
    def __init__(self, low, high):
        self.low = np.asarray(low, dtype=np.float32)
        self.high = np.asarray(high, dtype=np.float32)
        self.shape = self.low.shape
        self.dtype = np.float32


class _Dict:
    def __init__(self, spaces):
        self.spaces = dict(spaces)


REGISTRY = []


def install_stubs(shared_clients, client_class=None):
    """Put pybullet / pybullet_data / pybullet_utils.bullet_client / gym stubs in sys.modules.

    Every BulletClient() the reference constructs is appended to `shared_clients`.  client_class: FakeClient (default) or StaticWorldClient.
    """
    client_class = client_class or FakeClient
    module_client = client_class()

    pb = types.ModuleType('pybullet')
    for name in dir(FakeClient):
        if name.startswith('_'):
            continue
        attr = getattr(module_client, name)
        setattr(pb, name, attr)
    sys.modules['pybullet'] = pb

    pd = types.ModuleType('pybullet_data')
    pd.getDataPath = lambda: '<pybullet_data>'
    sys.modules['pybullet_data'] = pd

    pu = types.ModuleType('pybullet_utils')
    bc = types.ModuleType('pybullet_utils.bullet_client')

    def BulletClient(connection_mode=None):
        c = client_class(connection_mode)
        shared_clients.append(c)
        return c

    bc.BulletClient = BulletClient
    pu.bullet_client = bc
    sys.modules['pybullet_utils'] = pu
    sys.modules['pybullet_utils.bullet_client'] = bc

    gym = types.ModuleType('gym')
    gym.GoalEnv = type('GoalEnv', (object,), {})
    spaces = types.ModuleType('gym.spaces')
    spaces.Box = _Box
    spaces.Dict = _Dict
    utils = types.ModuleType('gym.utils')
    seeding = types.ModuleType('gym.utils.seeding')
    seeding.np_random = lambda seed=None: (np.random.RandomState(seed), seed)
    utils.seeding = seeding
    envs = types.ModuleType('gym.envs')
    registration = types.ModuleType('gym.envs.registration')

    def register(id, entry_point, **kw):
        REGISTRY.append({'id': id, 'entry_point': entry_point, 'kwargs': kw})

    registration.register = register
    envs.registration = registration
    gym.spaces, gym.utils, gym.envs = spaces, utils, envs
    for n, m in (('gym', gym), ('gym.spaces', spaces), ('gym.utils', utils), ('gym.utils.seeding', seeding),
                 ('gym.envs', envs), ('gym.envs.registration', registration)):
        sys.modules[n] = m
    return module_client
